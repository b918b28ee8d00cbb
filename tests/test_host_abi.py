"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol that
include/diner_hip.h declares; argument validation returns error codes (never aborts); the plug-in
class keeps the reference's constructor/attribute surface and holds no parameters or buffers."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "diner_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(diner_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from diner_amd import _lib
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/diner_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header drifted apart"
    # one ABI version in three places: the header's define, what the library was built with, what the ctypes table is written for
    header = (ROOT / "include" / "diner_hip.h").read_text()
    assert lib.diner_version() == _lib.ABI_VERSION == int(re.search(r"#define DINER_ABI_VERSION (\d+)", header).group(1))
    fp32_img = 16 * 7 * 256 + 13 * 16 * 64 * 256 + 64 * 256 + 14 * 512 + 32
    f16_img = (16 * 4 * 1024 + 13 * 16 * 32 * 1024) // 2 + 14 * 512 + 32 + 4 * 512   # halfs of 14 layers | biases | lin_out weights (fp32: VALU head)
    assert lib.diner_mlp_packed_floats() == fp32_img + f16_img


def test_argument_validation_returns_codes():
    from diner_amd import _lib
    lib = _lib.lib()
    # NULL pointers / bad sizes are rejected before any launch (no GPU is touched)
    assert lib.diner_composite(None, None, None, 4, 8, 1, None, None, None, None, None) == -1
    assert b"NULL" in lib.diner_last_error()
    assert lib.diner_composite(None, None, None, -1, 8, 1, None, None, None, None, None) == -1
    assert lib.diner_sample_coarse(None, 3, 0, None, 0, None, None) == -1
    assert lib.diner_pack_latent(None, 1, 512, 4, 4, None, None) == -1
    cfg = _lib.DinerSamplerCfg(10, 4, 5, 0.05)  # n_gaussian > n_samples (nerf_renderer.py:89)
    sc = _lib.DinerScene()
    sc.SB, sc.NV, sc.H, sc.W, sc.image_w, sc.image_h = 1, 1, 2, 2, 2.0, 2.0
    sc.poses = sc.focal = sc.c = sc.maps = 8  # non-NULL dummies, never dereferenced
    assert lib.diner_sample_depthguided(C.byref(sc), None, 0, C.byref(cfg), None, None, None, None, 0, None, None, None, None) == -1
    assert b"n_gaussian" in lib.diner_last_error()
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(NotImplementedError):
        _lib.check(-3, "x")


def test_plugin_surface_matches_reference():
    import inspect

    import diner_amd
    r = diner_amd.NeRFRendererDGS()
    # reference defaults (src/models/nerf_renderer.py:23-37)
    assert (r.n_samples, r.n_depth_candidates, r.n_gaussian, r.eval_batch_size, r.white_bkgd) == (40, 1000, 15, 100000, True)
    r2 = diner_amd.NeRFRendererDGS(n_samples=64, n_depth_candidates=500, n_gaussian=24, white_bkgd=False)
    r2.n_samples, r2.n_gaussian = 128, int(24 * 128 / 64)  # python_scripts/create_prediction_folder.py:49-52
    assert isinstance(r, torch.nn.Module) and list(r.state_dict()) == [] and list(r.parameters()) == []
    sig = inspect.signature(r.forward)
    assert list(sig.parameters)[:3] == ["model", "rays", "want_weights"] and sig.parameters["want_weights"].default is False
    for name in ("sample_coarse", "sample_depthguided", "fill_up_uniform_samples", "composite", "render_rays"):
        assert callable(getattr(r, name))
    # dotted-path resolution as the reference's import_obj does it (src/util/import_helper.py:4-24)
    import importlib
    assert importlib.import_module("diner_amd").__dict__["NeRFRendererDGS"] is diner_amd.NeRFRendererDGS


def test_cpu_rays_are_rejected_loudly():
    import diner_amd
    r = diner_amd.NeRFRendererDGS(n_samples=8, n_depth_candidates=16, n_gaussian=2)
    with pytest.raises(RuntimeError, match="GPU only"):
        r._check_rays(torch.zeros(1, 4, 8))
    with pytest.raises(AssertionError):
        r.forward(None, torch.zeros(4, 8))  # rank-3 rays required (nerf_renderer.py:412)
