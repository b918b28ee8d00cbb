"""The default arithmetic (f16x3: fp32 operands split into fp16 hi+lo) against magnitude.

fp16 has an absolute floor (subnormals, 2^-24) and a ceiling (65504; the hidden state is carried at 2^-4, so an
activation overflows at |x| >= ~1.05e6).  Real ResNet34 latents and trained weights are not unit-scale like the
synthetic ones, so: sweep latent scale x weight scale x bias scale, and require for every cell EITHER fp32-grade
agreement with the exact-fp32-MFMA kernel and the CPU oracle (the same bars as tests/test_gpu_parity.py) OR a loud
failure -- non-finite samples that `forward()` turns into a RuntimeError naming precision='fp32' -- never finite
garbage.  The measured envelope is quoted in DESIGN.md §2."""
import numpy as np
import pytest
import torch

from synthetic import synth

pytestmark = pytest.mark.gpu

LATENT = [0.01, 1.0, 30.0]
WEIGHT = [0.3, 1.0, 3.0]
BIAS = [0.0, 1.0]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _case(ls, ws, bs):
    sc = synth.make_scene(24, 24, 3, seed=5, feature_padding=4, latent_scale=ls)
    w = synth.make_mlp_weights(7, bias_scale=bs)    # seed 7: sigma > 0 on most samples (bench.py uses it for the same reason)
    w = {k: (v * np.float32(ws) if k.endswith("weight") else v) for k, v in w.items()}
    rays = sc.target_rays()[:, ::7]
    return sc, w, rays


@pytest.mark.parametrize("bs", BIAS)
@pytest.mark.parametrize("ws", WEIGHT)
@pytest.mark.parametrize("ls", LATENT)
def test_f16x3_envelope(ls, ws, bs):
    sc, w, rays = _case(ls, ws, bs)
    _fp32_grade_or_loud(sc, w, rays, f"latent x{ls:g} weights x{ws:g} bias {bs:g}")


def _cnn_case(gain, seed=11):
    """Statistics a trained ResNet34 trunk + MLP actually has, unlike the unit-scale gaussians above: the
    latent is POST-ReLU (non-negative, ~half of it exactly 0), HEAVY-TAILED (half-Student-t, 3 degrees of freedom: outliers of
    30-100 sigma) with PER-CHANNEL scales spread log-uniformly over 1e-3 ... 1e2; every weight matrix has its output rows scaled
    log-uniformly over a factor 10 (row norms of trained layers are far from equal).  `gain` scales the whole latent."""
    sc, w, rays = _case(1.0, 1.0, 1.0)
    rs = np.random.RandomState(seed)
    shape = sc.latent.shape                                   # [1, NV, C, h, w]
    t = rs.standard_t(3, size=shape).astype(np.float32)
    ch = np.exp(rs.uniform(np.log(1e-3), np.log(1e2), size=(1, 1, shape[2], 1, 1))).astype(np.float32)
    sc.latent = np.ascontiguousarray(np.maximum(t, 0.0) * ch * np.float32(gain))
    out = {}
    for k, v in w.items():
        if k.endswith("weight") and v.ndim == 2:
            rows = np.exp(rs.uniform(-0.5 * np.log(10.0), 0.5 * np.log(10.0), size=(v.shape[0], 1))).astype(np.float32)
            v = (v * rows).astype(np.float32)
        out[k] = v
    return sc, out, rays    # (at gain 1 the latent has std 25 and outliers beyond 2000: the three gains span std 0.75 ... 750)


@pytest.mark.parametrize("gain", [0.03, 1.0, 30.0])
def test_f16x3_envelope_cnn_statistics(gain):
    sc, w, rays = _cnn_case(gain)
    lat = sc.latent
    assert float(lat.min()) >= 0.0 and 0.3 < float((lat == 0).mean()) < 0.7 and float(lat.max()) > 50 * float(lat.std())
    _fp32_grade_or_loud(sc, w, rays, f"CNN-like latent (post-ReLU, heavy-tailed, channel scales 1e-3..1e2) x{gain:g}, weight rows x10 spread")


def _fp32_grade_or_loud(sc, w, rays, label):
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    from oracle.oracle import Oracle
    dev = torch.device("cuda:0")
    K = 16
    m = model_from_scene(sc, w, device=dev)
    z = np.sort(np.random.RandomState(1).uniform(sc.near, sc.far, (1, rays.shape[1], K)).astype(np.float32), -1)
    xyz = rays[0, :, None, :3] + z[0, ..., None] * rays[0, :, None, 3:6]
    dirs = np.broadcast_to(rays[0, :, None, 3:6], xyz.shape)
    ref = Oracle(sc, w).points_forward(xyz.reshape(-1, 3), dirs.reshape(-1, 3)).reshape(z.shape[1], K, 4)
    out = {}
    for prec in ("fp32", "f16x3"):
        r = NeRFRendererDGS(n_samples=K, n_depth_candidates=64, n_gaussian=4, white_bkgd=True)
        r.precision = prec
        with torch.no_grad():
            out[prec] = r.render_points(m, T(rays, dev), T(z, dev)).cpu().numpy()[0]
    finite_ref = np.isfinite(ref).all()
    # "fp32-grade": the north_star bars (1e-4 on rgb, 1e-4 * max(1, sigma/12) on sigma) OR 3x the disagreement of two
    # legitimate fp32 evaluations of the same network (exact-fp32-MFMA kernel vs the CPU oracle's fma chains), whichever is
    # larger -- away from unit scale fp32 itself is not accurate to 1e-4 absolute (activations x30 -> rounding noise x30)
    e32 = np.abs(out["fp32"] - ref) if finite_ref else None
    f16 = out["f16x3"]

    def check(got):
        e = np.abs(got - ref)
        rgb_tol = max(1e-4, 3 * float(e32[..., :3].max()))
        sig_tol = np.maximum(1e-4 * np.maximum(1.0, ref[..., 3] / 12.0), 3 * float(e32[..., 3].max()))
        assert float(e[..., :3].max()) <= rgb_tol, (float(e[..., :3].max()), rgb_tol)
        assert np.all(e[..., 3] <= sig_tol), (float(e[..., 3].max()), float(e32[..., 3].max()))
        return e

    if np.isfinite(f16).all():
        assert finite_ref
        e16 = check(f16)
        verdict = (f"fp32-grade: |f16x3-oracle| rgb {e16[..., :3].max():.1e} sigma {e16[..., 3].max():.1e}; "
                   f"|fp32-oracle| rgb {e32[..., :3].max():.1e} sigma {e32[..., 3].max():.1e}")
    else:
        # outside the envelope: must be loud.  forward() raises (deferred check -> check_finite()), fp32 mode renders it.
        r = NeRFRendererDGS(n_samples=K, n_depth_candidates=64, n_gaussian=4, white_bkgd=True)
        with torch.no_grad():
            r(m, T(rays, dev), z_samples=T(z, dev))
            with pytest.raises(RuntimeError, match="precision = 'fp32'"):
                r.check_finite()
            r.precision = "fp32"
            o = r(m, T(rays, dev), z_samples=T(z, dev))
            if finite_ref:
                r.check_finite()
                assert bool(torch.isfinite(o.fine.rgb).all())
            else:            # fp32 overflows on this cell too: examined here, so that no unexamined frame outlives the test
                with pytest.raises(RuntimeError, match="the model itself"):
                    r.check_finite()
        # non-finite samples are NaN/inf, never finite garbage next to a finite oracle value
        bad = ~np.isfinite(f16).all(-1)
        if finite_ref and (~bad).any():
            e = np.abs(f16 - ref)[~bad]
            assert float(e.max()) <= max(1e-3, 30 * float(e32.max())), "finite garbage next to non-finite samples"
        verdict = f"LOUD ({bad.mean():.0%} of the samples non-finite)"
    print(f"{label}: sigma max {np.nanmax(ref[..., 3]):.3g} -> f16x3 {verdict}")


def test_nonfinite_is_raised_not_returned():
    """A latent far outside the fp16 range: f16x3 must raise through every entry (deferred, sync and render_image)."""
    from diner_amd import NeRFRendererDGS
    from synthetic.model_stub import model_from_scene
    dev = torch.device("cuda:0")
    sc, w, rays = _case(1e7, 1.0, 0.0)
    m = model_from_scene(sc, w, device=dev)
    r = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
    with torch.no_grad():
        r(m, T(rays, dev))                       # deferred: the call itself returns
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="inf/NaN"):
            r(m, T(rays, dev))                   # ... the next one reports it
        r.finite_check = "sync"
        with pytest.raises(RuntimeError, match="inf/NaN"):
            r(m, T(rays, dev))
        r.finite_check = "off"
        o = r(m, T(rays, dev))
        assert not bool(torch.isfinite(o.fine.rgb).all())
        r.finite_check = "deferred"
        E = T(sc.target_extrinsics[None], dev)
        Kt = T(sc.target_intrinsics[None], dev)
        with pytest.raises(RuntimeError, match="inf/NaN"):
            r.render_image(m, E, Kt, 8, 8, sc.near, sc.far)
        # a healthy model afterwards is not blamed for the old frames
        sc2, w2, rays2 = _case(1.0, 1.0, 0.0)
        m2 = model_from_scene(sc2, w2, device=dev)
        r(m2, T(rays2, dev))
        r.check_finite()


# ---- the non-finite guard leaves no window -------------------------------------------------------------------
_LAST_CALL_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.test_gpu_magnitude import _case, T
from diner_amd import NeRFRendererDGS
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
sc, w, rays = _case(%s, 1.0, 0.0)
m = model_from_scene(sc, w, device=dev)
sc0, w0, rays0 = _case(1.0, 1.0, 0.0)
m0 = model_from_scene(sc0, w0, device=dev)          # a healthy model for the calls before the last
r = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
assert rays.shape[1] <= r.finite_sync_rays      # small chunks: the deferred path
with torch.no_grad():
    for i in range(3):
        r(m0, T(rays0, dev))
    o = r(m, T(rays, dev))                      # the LAST call of the program; nobody calls check_finite()
print("frames rendered", flush=True)
"""


@pytest.mark.parametrize("latent_scale,bad", [("1e7", True), ("1.0", False)])
def test_last_call_of_a_program_is_examined(latent_scale, bad):
    """forward() cannot end a program with an unexamined non-finite frame: a script whose LAST call is out of the f16x3 envelope
    and which never calls check_finite() ends with a non-zero status and the RuntimeError text on stderr (weakref.finalize of the
    renderer at interpreter exit); the same script inside the envelope exits 0."""
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parents[1])
    p = subprocess.run([sys.executable, "-c", _LAST_CALL_SCRIPT % (root, latent_scale)], capture_output=True, text=True, timeout=600, cwd=root)
    assert "frames rendered" in p.stdout, p.stderr[-2000:]
    if bad:
        assert p.returncode == 70 and "inf/NaN" in p.stderr and "precision = 'fp32'" in p.stderr, (p.returncode, p.stderr[-2000:])
    else:
        assert p.returncode == 0, p.stderr[-2000:]


def test_nonfinite_window_is_closed_in_process():
    from diner_amd import NeRFRendererDGS
    from diner_amd.renderer import _FiniteGuard
    from synthetic.model_stub import model_from_scene
    import gc
    dev = torch.device("cuda:0")
    sc, w, rays = _case(1e7, 1.0, 0.0)
    m = model_from_scene(sc, w, device=dev)
    good_sc, good_w, good_rays = _case(1.0, 1.0, 0.0)
    good = model_from_scene(good_sc, good_w, device=dev)
    with torch.no_grad():
        # (1) a call that carries more than the reference's 4096-ray chunk is examined before it returns
        r = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
        big = T(np.repeat(rays, 4096 // rays.shape[1] + 1, axis=1), dev)
        assert big.shape[1] > r.finite_sync_rays
        with pytest.raises(RuntimeError, match="inf/NaN"):
            r(m, big)
        # (2) ... and so is one that returns the weights
        with pytest.raises(RuntimeError, match="inf/NaN"):
            r(m, T(rays, dev), want_weights=True)
        # (3) small chunks: the host never waits for the chunk it has just issued, and never has more than two unexamined
        r = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
        for i in range(6):
            r(good, T(good_rays, dev))
            assert len(r._guard.pending) <= 2
        r.check_finite()
        # (4) a renderer that is dropped with an unexamined bad chunk: the finding is raised by the next call of ANY renderer
        r = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
        r(m, T(rays, dev))
        del r
        gc.collect()
        r2 = NeRFRendererDGS(n_samples=16, n_depth_candidates=64, n_gaussian=4)
        with pytest.raises(RuntimeError, match="earlier renderer was collected"):
            r2(good, T(good_rays, dev))
        assert not _FiniteGuard.unreported
        r2(good, T(good_rays, dev))
        r2.check_finite()
