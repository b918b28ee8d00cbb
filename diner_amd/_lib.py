"""ctypes binding of ``libdiner_hip.so`` (the C ABI declared in ``include/diner_hip.h``).

There is NO fallback: if the HIP library is missing the import of the product path fails loudly
(build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C diner_amd/csrc``).
``torch`` is imported first on purpose: its bundled ``libamdhip64.so`` (soname ``libamdhip64.so.7``)
is then the HIP runtime our library binds to, so torch's streams and device pointers are valid
inside our launches.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch  # noqa: F401  (must be loaded before libdiner_hip.so, see above)

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libdiner_hip.so"
_FP = C.c_void_p  # device pointers travel as integers

N_BLOCKS, COMBINE = 5, 3
# the ABI version THIS binding (the argument lists in SYMBOLS below) is written against = DINER_ABI_VERSION of include/diner_hip.h
ABI_VERSION = 2
PRECISIONS = {"fp32": 0, "f16x3": 1}


class DinerScene(C.Structure):
    _fields_ = [("SB", C.c_int32), ("NV", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32), ("C", C.c_int32), ("num_freqs", C.c_int32),
                ("image_w", C.c_float), ("image_h", C.c_float), ("feature_padding", C.c_float),
                ("freq_factor", C.c_float),
                ("poses", _FP), ("focal", _FP), ("c", _FP), ("maps", _FP), ("latent", _FP), ("linz_maps", _FP)]


class DinerMlpRaw(C.Structure):
    _fields_ = [("lin_in_w", _FP), ("lin_in_b", _FP),
                ("lin_z_w", _FP * COMBINE), ("lin_z_b", _FP * COMBINE),
                ("fc0_w", _FP * N_BLOCKS), ("fc0_b", _FP * N_BLOCKS),
                ("fc1_w", _FP * N_BLOCKS), ("fc1_b", _FP * N_BLOCKS),
                ("lin_out_w", _FP), ("lin_out_b", _FP)]


class DinerTargetCam(C.Structure):
    _fields_ = [("extrinsics", _FP), ("intrinsics", _FP), ("z_near", _FP), ("z_far", _FP), ("H", C.c_int32), ("W", C.c_int32)]


class DinerSamplerCfg(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("n_samples", C.c_int32), ("n_gaussian", C.c_int32),
                ("depth_diff_max", C.c_float)]


# every symbol include/diner_hip.h declares: name -> (restype, argtypes)
_I64, _I32, _U64, _P = C.c_int64, C.c_int32, C.c_uint64, C.c_void_p
SYMBOLS = {
    "diner_last_error": (C.c_char_p, []),
    "diner_version": (C.c_int, []),
    "diner_gen_rays": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "diner_depth2normal": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "diner_pack_maps": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "diner_pack_maps_from_depth": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "diner_pack_latent": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "diner_mlp_packed_floats": (_I64, []),
    "diner_pack_mlp": (C.c_int, [C.POINTER(DinerMlpRaw), _P, _P]),
    "diner_pack_linz_maps": (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _P]),
    "diner_sample_coarse": (C.c_int, [_P, _I64, _I32, _P, _U64, _P, _P]),
    "diner_sample_depthguided": (C.c_int, [C.POINTER(DinerScene), _P, _I64, C.POINTER(DinerSamplerCfg),
                                           _P, _P, _P, _P, _U64, _P, _P, _P, _P]),
    "diner_fill_up_uniform_samples": (C.c_int, [_P, _P, _I64, _I32, _P, _U64, _P, _P]),
    "diner_render_points_scratch_floats": (_I64, [_I64, _I32, _I32]),
    "diner_render_points": (C.c_int, [C.POINTER(DinerScene), _P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "diner_composite": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "diner_decode_depth_u16": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         _P, _P, _P, _P]),
    "diner_train_gemm": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I64,
                                   _I32, _P, _P, _I32, _I32, _P]),
    "diner_train_amax": (C.c_int, [_P, _I64, _P, _P]),
    "diner_train_colsum_amax": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _P]),
    "diner_train_split_panel": (C.c_int, [_P, _I32, _I64, _I32, _I32, _P, _P, _P]),
    "diner_train_pack_core": (C.c_int, [_P, _I64, _I32, _I32, _P, _P]),
    "diner_train_gemm_core": (C.c_int, [_P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _I64, _I64, _I32, _P, _I32, _I32, _P, _P, _P]),
    "diner_train_gemm_panel": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I32, _I32, _P]),
    "diner_train_colsum": (C.c_int, [_P, _I64, _I32, _I64, _P, _P]),
    "diner_train_point_inputs": (C.c_int, [C.POINTER(DinerScene), _P, _I32, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "diner_train_bilinear_scatter": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "diner_train_nhwc_to_nchw": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "diner_train_view_mean": (C.c_int, [_P, _I64, _I32, _P, _I32, _P]),
    "diner_train_head": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _P]),
    "diner_composite_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "diner_render_workspace_floats": (_I64, [_I64, _I64, _I32, _I32, _I32]),
    "diner_render_image_workspace_floats": (_I64, [_I64, _I32, _I32, _I32, _I32, _I32]),
    "diner_render_image": (C.c_int, [C.POINTER(DinerScene), _P, C.POINTER(DinerTargetCam), C.POINTER(DinerSamplerCfg), _I32, _I32, _U64,
                                     _P, _P, _P, _P, _P, _P, _P]),
    "diner_render": (C.c_int, [C.POINTER(DinerScene), _P, _P, _I64, C.POINTER(DinerSamplerCfg), _I32, _I32,
                               _P, _P, _P, _U64, _P, _P, _P, _P, _P, _P]),
}

_lib = None


class DinerHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The loaded library with typed entry points; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the MI355X render path has no CPU or PyTorch fallback. "
                "Build it with `make -C diner_amd/csrc` (hipcc --offload-arch=gfx950).")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the ABI and the header drift apart
            fn.restype, fn.argtypes = res, args
        if l.diner_version() != ABI_VERSION:   # a stale .so (e.g. a git-ignored build from before an ABI change): never call into it
            raise ImportError(f"{LIB_PATH} has ABI version {l.diner_version()}, this binding needs {ABI_VERSION}: rebuild it "
                              "(`make -C diner_amd/csrc`)")
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().diner_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        if rc == -3:
            raise NotImplementedError(f"{what}: {msg}")
        raise DinerHipError(f"{what}: {msg} (rc={rc})")
