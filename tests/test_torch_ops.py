"""torch operator registration (diner_amd/csrc/torch_ops.cpp, diner_amd/ops.py): the extension builds against the installed
torch, registers ``diner::render`` / ``diner::render_image``, refuses host tensors, and -- on the GPU -- gives bit-identical
frames to the ctypes binding (both call the same C ABI)."""
import numpy as np
import pytest
import torch


def test_extension_builds_and_registers():
    from diner_amd import ops
    ops.build()
    o = ops.load()
    for name in ("render", "render_image"):
        schema = str(getattr(o, name).default._schema)
        assert schema.startswith(f"diner::{name}(Tensor maps, Tensor poses"), schema
    assert "Tensor? status" in str(o.render.default._schema)
    from diner_amd import _lib
    assert int(o.abi_version()) == _lib.lib().diner_version() == _lib.ABI_VERSION
    # no CPU kernel is registered: host tensors are refused by the dispatcher, not silently computed elsewhere
    t = torch.zeros(1)
    with pytest.raises((NotImplementedError, RuntimeError)):
        o.render(t, t, t, t, t, None, t, t, 1.0, 1.0, 0.0, 6, 1.0, 8, 4, 1, 0.05, True, 1, 0, False, None)


def test_stale_extension_is_not_used(tmp_path, monkeypatch):
    """an extension built against another libdiner_hip.so / torch_ops.cpp / header may carry another argument list.
    It is recognised by content (the stamp written by build()), not taken as a binding, and load() refuses it."""
    from diner_amd import ops
    ops.build()
    assert ops.available() and not ops.stale()
    good = ops.STAMP.read_text()
    try:
        ops.STAMP.write_text("0" * 64 + "\n")       # = built from other sources
        assert ops.stale() and not ops.available()
        import diner_amd
        assert diner_amd.NeRFRendererDGS().binding == "ctypes"
        monkeypatch.setattr(ops, "_loaded", False)
        with pytest.raises(RuntimeError, match="rebuild"):
            ops.load()
    finally:
        ops.STAMP.write_text(good)
    assert ops.available()


@pytest.mark.gpu
def test_torch_ops_binding_equals_ctypes_binding():
    from diner_amd import NeRFRendererDGS
    from synthetic import synth
    from synthetic.model_stub import model_from_scene
    dev = torch.device("cuda:0")
    sc = synth.make_scene(24, 32, 3, seed=5, feature_padding=4)
    m = model_from_scene(sc, synth.make_mlp_weights(6, bias_scale=0.1), device=dev)
    rays = torch.from_numpy(sc.target_rays()[:, ::3]).to(dev)
    H, W = 12, 20
    E = torch.from_numpy(np.ascontiguousarray(sc.target_extrinsics, dtype=np.float32))[None].to(dev)
    Kt = torch.tensor([[[1.2 * W, 0, W / 2], [0, 1.2 * W, H / 2], [0, 0, 1]]], dtype=torch.float32, device=dev)
    outs = {}
    for binding in ("torch_ops", "ctypes"):
        for precision in ("f16x3", "fp32"):
            r = NeRFRendererDGS(n_samples=16, n_depth_candidates=128, n_gaussian=5, white_bkgd=sc.white_bkgd)
            assert r.binding == "torch_ops"          # the default once the extension is built
            r.binding, r.precision, r.seed = binding, precision, 11
            with torch.no_grad():
                o = r(m, rays, want_weights=True).fine
                img, dep = r.render_image(m, E, Kt, H, W, float(sc.near), float(sc.far), return_depth=True)
            outs[binding, precision] = (o.rgb, o.depth, o.weights, img, dep)
    for precision in ("f16x3", "fp32"):
        for a, b in zip(outs["torch_ops", precision], outs["ctypes", precision]):
            assert torch.equal(a, b)
    # the non-finite status word travels through the op as well
    r = NeRFRendererDGS(n_samples=16, n_depth_candidates=128, n_gaussian=5, white_bkgd=sc.white_bkgd)
    r.finite_check = "sync"
    m.mlp_fine.lin_out.bias.data[0] = float("nan")
    with pytest.raises(RuntimeError, match="inf/NaN"):
        with torch.no_grad():
            r(m, rays)
