"""ctypes front-end of the CPU oracle (``oracle/diner_oracle.c``).

TEST INFRASTRUCTURE: imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package ``diner_amd``.
All arrays are numpy float32, C-contiguous, in the reference's layouts (single scene).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libdiner_oracle.so"
_FP = C.POINTER(C.c_float)
N_BLOCKS, COMBINE = 5, 3


class _Scene(C.Structure):
    _fields_ = [("NV", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("C", C.c_int32),
                ("poses", _FP), ("focal", _FP), ("c", _FP),
                ("image_w", C.c_float), ("image_h", C.c_float),
                ("depths", _FP), ("depths_std", _FP), ("normals", _FP), ("latent", _FP),
                ("feature_padding", C.c_float), ("freq_factor", C.c_float)]


class _Mlp(C.Structure):
    _fields_ = [("lin_in_w", _FP), ("lin_in_b", _FP),
                ("lin_z_w", _FP * COMBINE), ("lin_z_b", _FP * COMBINE),
                ("fc0_w", _FP * N_BLOCKS), ("fc0_b", _FP * N_BLOCKS),
                ("fc1_w", _FP * N_BLOCKS), ("fc1_b", _FP * N_BLOCKS),
                ("lin_out_w", _FP), ("lin_out_b", _FP)]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (a few seconds)."""
    src = _HERE / "diner_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        # DINER_ORACLE_SO: another build of the same file, e.g. the sanitizer build of `make -C oracle asan-test`
        _lib = C.CDLL(os.environ.get("DINER_ORACLE_SO") or str(build()))
        _lib.orc_mlp_prepare.restype = C.c_void_p
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_FP)


class Oracle:
    """Holds a scene + MLP weights and exposes the stage functions of the restatement."""

    def __init__(self, scene, weights, freq_factor: float = 6.28, threads: int | None = None):
        if threads is not None:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        self._keep = []
        self.NV, self.H, self.W = scene.NV, scene.H, scene.W
        s = _Scene()
        s.NV, s.H, s.W = scene.NV, scene.H, scene.W
        s.image_w, s.image_h = float(scene.image_shape[0]), float(scene.image_shape[1])
        s.feature_padding, s.freq_factor = float(scene.feature_padding), float(freq_factor)
        for name in ("poses", "focal", "c", "depths", "depths_std", "normals"):
            arr, p = _f(getattr(scene, name)[0])
            self._keep.append(arr)
            setattr(s, name, p)
        if scene.latent is not None:
            lat, p = _f(scene.latent[0])
            self._keep.append(lat)
            s.latent = p
            s.C, s.h, s.w = lat.shape[1], lat.shape[2], lat.shape[3]
        self._scene = s
        self._mlp_t = None
        if weights is not None:
            m = _Mlp()

            def put(field, key, idx=None):
                arr, p = _f(weights[key])
                self._keep.append(arr)
                if idx is None:
                    setattr(m, field, p)
                else:
                    getattr(m, field)[idx] = p

            put("lin_in_w", "lin_in.weight"), put("lin_in_b", "lin_in.bias")
            put("lin_out_w", "lin_out.weight"), put("lin_out_b", "lin_out.bias")
            for b in range(COMBINE):
                put("lin_z_w", f"lin_z.{b}.weight", b), put("lin_z_b", f"lin_z.{b}.bias", b)
            for b in range(N_BLOCKS):
                put("fc0_w", f"blocks.{b}.fc_0.weight", b), put("fc0_b", f"blocks.{b}.fc_0.bias", b)
                put("fc1_w", f"blocks.{b}.fc_1.weight", b), put("fc1_b", f"blocks.{b}.fc_1.bias", b)
            self._mlp = m
            self._mlp_t = C.c_void_p(lib().orc_mlp_prepare(C.byref(m)))

    def __del__(self):
        if getattr(self, "_mlp_t", None):
            lib().orc_mlp_free(self._mlp_t)
            self._mlp_t = None

    # ---- stages -------------------------------------------------------------------------
    def sample_coarse(self, rays, NC, u_coarse):
        rays, pr = _f(rays.reshape(-1, 8))
        u, pu = _f(u_coarse)
        z = np.empty((rays.shape[0], NC), np.float32)
        lib().orc_sample_coarse(pr, C.c_int64(rays.shape[0]), C.c_int(NC), pu, z.ctypes.data_as(_FP))
        return z

    def likelihood(self, rays, z_cand):
        rays, pr = _f(rays.reshape(-1, 8))
        zc, pz = _f(z_cand)
        L = np.empty_like(zc)
        lib().orc_likelihood(C.byref(self._scene), pr, C.c_int64(rays.shape[0]), C.c_int(zc.shape[1]),
                             pz, L.ctypes.data_as(_FP))
        return L

    def sample_depthguided(self, rays, z_cand, K, G, n_gauss, want_L=False):
        rays, pr = _f(rays.reshape(-1, 8))
        zc, pz = _f(z_cand)
        ng, pn = _f(n_gauss if G > 0 else np.zeros((rays.shape[0], 1)))
        z = np.empty((rays.shape[0], K), np.float32)
        L = np.empty_like(zc) if want_L else None
        lib().orc_sample_depthguided(C.byref(self._scene), pr, C.c_int64(rays.shape[0]),
                                     C.c_int(zc.shape[1]), C.c_int(K), C.c_int(G), pz, pn,
                                     z.ctypes.data_as(_FP), L.ctypes.data_as(_FP) if want_L else None)
        return (z, L) if want_L else z

    def fill_up(self, rays, z, u_fill):
        rays, pr = _f(rays.reshape(-1, 8))
        z, pz = _f(z)
        u, pu = _f(u_fill)
        out = np.empty_like(z)
        lib().orc_fill_up(pr, C.c_int64(rays.shape[0]), C.c_int(z.shape[1]), pz, pu, out.ctypes.data_as(_FP))
        return out

    def point_inputs(self, xyz, dirs):
        xyz, px = _f(xyz.reshape(-1, 3))
        dirs, pd = _f(dirs.reshape(-1, 3))
        out = np.empty((self.NV, xyz.shape[0], self._scene.C + 55), np.float32)
        lib().orc_point_inputs(C.byref(self._scene), px, pd, C.c_int64(xyz.shape[0]), out.ctypes.data_as(_FP))
        return out

    def mlp_forward(self, mlp_input):
        x, px = _f(mlp_input)
        out = np.empty((x.shape[1], 4), np.float32)
        lib().orc_mlp_forward(self._mlp_t, px, C.c_int(x.shape[0]), C.c_int64(x.shape[1]), out.ctypes.data_as(_FP))
        return out

    def points_forward(self, xyz, dirs):
        xyz, px = _f(xyz.reshape(-1, 3))
        dirs, pd = _f(dirs.reshape(-1, 3))
        out = np.empty((xyz.shape[0], 4), np.float32)
        lib().orc_points_forward(C.byref(self._scene), self._mlp_t, px, pd, C.c_int64(xyz.shape[0]),
                                 out.ctypes.data_as(_FP))
        return out

    def composite(self, rays, z, rgbsigma, white_bkgd=True):
        rays, pr = _f(rays.reshape(-1, 8))
        z, pz = _f(z)
        c, pc = _f(rgbsigma)
        NR, K = z.shape
        w, rgb, depth = np.empty((NR, K), np.float32), np.empty((NR, 3), np.float32), np.empty(NR, np.float32)
        lib().orc_composite(pr, pz, pc, C.c_int64(NR), C.c_int(K), C.c_int(int(white_bkgd)),
                            w.ctypes.data_as(_FP), rgb.ctypes.data_as(_FP), depth.ctypes.data_as(_FP))
        return w, rgb, depth

    def render(self, rays, NC, K, G, noise, white_bkgd=True):
        """Whole ``NeRFRendererDGS.forward`` with dense noise; returns a dict of stage outputs."""
        rays, pr = _f(rays.reshape(-1, 8))
        NR = rays.shape[0]
        (u, pu), (n, pn), (uf, puf) = _f(noise[0]), _f(noise[1] if G > 0 else np.zeros((NR, 1))), _f(noise[2])
        z = np.empty((NR, K), np.float32)
        c = np.empty((NR, K, 4), np.float32)
        w, rgb, depth = np.empty((NR, K), np.float32), np.empty((NR, 3), np.float32), np.empty(NR, np.float32)
        lib().orc_render(C.byref(self._scene), self._mlp_t, pr, C.c_int64(NR), C.c_int(NC), C.c_int(K),
                         C.c_int(G), C.c_int(int(white_bkgd)), pu, pn, puf, z.ctypes.data_as(_FP),
                         c.ctypes.data_as(_FP), w.ctypes.data_as(_FP), rgb.ctypes.data_as(_FP),
                         depth.ctypes.data_as(_FP))
        return dict(z=z, rgbsigma=c, weights=w, rgb=rgb, depth=depth)
