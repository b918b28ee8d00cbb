"""The adjacent per-image producers: gen_rays and depth2normal.
CPU: the numpy restatements in synthetic/synth.py against goldens from the unmodified reference.
GPU: the HIP kernels (C ABI) against the same goldens."""
import json

import numpy as np
import pytest

from synthetic import synth
from oracle.gen_golden import digest, glue_inputs
from tests.conftest import GOLDEN_DIR


@pytest.fixture(scope="module")
def glue():
    g = glue_inputs()
    gold = dict(np.load(GOLDEN_DIR / "glue.npz", allow_pickle=False))
    want = json.loads(str(gold["digests"]))["inputs"]
    assert digest(*[g[k] for k in ("extrinsics", "intrinsics", "z_near", "z_far", "dmap")]) == want
    return g, gold


def test_numpy_restatements_match_reference(glue):
    g, gold = glue
    for b in range(2):
        r = synth.gen_rays(g["extrinsics"][b], g["intrinsics"][b], g["W"], g["H"], g["z_near"][b], g["z_far"][b])
        np.testing.assert_allclose(r, gold["rays"][b], rtol=0, atol=3e-7)
    n = synth.depth2normal(g["dmap"], g["intrinsics"])
    assert np.array_equal(np.isnan(n), np.isnan(gold["normals"]))
    np.testing.assert_allclose(np.nan_to_num(n), np.nan_to_num(gold["normals"]), rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_hip_gen_rays_and_depth2normal(glue):
    import torch
    from diner_amd import glue as hip
    g, gold = glue
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rays = hip.gen_rays(t(g["extrinsics"]), t(g["intrinsics"]), g["W"], g["H"], t(g["z_near"]), t(g["z_far"])).cpu().numpy()
    assert rays.shape == gold["rays"].shape
    np.testing.assert_allclose(rays, gold["rays"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(np.linalg.norm(rays[..., 3:6], axis=-1), 1.0, atol=1e-6)
    n = hip.depth2normal(t(g["dmap"]), t(g["intrinsics"])).cpu().numpy()
    assert np.array_equal(np.isnan(n), np.isnan(gold["normals"]))
    # unit normals from differences of nearby points: the cross product amplifies 1-ulp differences of the
    # re-projected points (fma contraction in ATen differs per op), hence 2e-5 rather than 1e-6
    np.testing.assert_allclose(np.nan_to_num(n), np.nan_to_num(gold["normals"]), rtol=0, atol=2e-5)
    bg = g["dmap"][:, 0] == 0
    assert np.all(n.transpose(0, 2, 3, 1)[bg] == 0)


@pytest.mark.gpu
def test_fused_depth2normal_packing_equals_two_step(glue):
    """diner_pack_maps_from_depth == depth2normal followed by diner_pack_maps (bit for bit)."""
    import ctypes as C

    import torch
    from diner_amd import _lib
    from diner_amd import glue as hip
    g, _ = glue
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = t(g["dmap"])[None]                                            # [1,N,1,H,W]
    std = torch.rand_like(d) * 0.01
    k = t(g["intrinsics"])[None]
    fused = hip.pack_maps_from_depth(d, std, k)
    n = hip.depth2normal(d[0], k[0])
    N, _, H, W = d[0].shape
    two = torch.empty((1, N, H, W, 8), device=dev)
    _lib.check(_lib.lib().diner_pack_maps(d.data_ptr(), std.data_ptr(), n.data_ptr(), N, H, W, two.data_ptr(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "diner_pack_maps")
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(fused), torch.nan_to_num(two))


@pytest.mark.gpu
def test_render_image_equals_chunked_forward():
    """``render_image`` (rays generated on the GPU, one launch) against the reference's flow of
    predict_imgs_from_batch (src/models/diner.py:75-97): gen_rays -> forward on the ray tensor -> view/permute."""
    import torch
    from diner_amd import NeRFRendererDGS, glue
    from synthetic import synth
    from synthetic.model_stub import model_from_scene
    dev = torch.device("cuda:0")
    sc = synth.make_scene(24, 32, 3, seed=5, feature_padding=4)
    m = model_from_scene(sc, synth.make_mlp_weights(6, bias_scale=0.1), device=dev)
    r = NeRFRendererDGS(n_samples=16, n_depth_candidates=128, n_gaussian=5, white_bkgd=sc.white_bkgd)
    H, W = 20, 28
    E = torch.from_numpy(np.ascontiguousarray(sc.target_extrinsics, dtype=np.float32))[None].to(dev)
    Kt = torch.tensor([[[1.2 * W, 0, W / 2], [0, 1.2 * W, H / 2], [0, 0, 1]]], dtype=torch.float32, device=dev)
    near, far = float(sc.near), float(sc.far)
    r.seed, r._calls = 3, 0
    rgb, depth = r.render_image(m, E, Kt, H, W, near, far, return_depth=True)
    assert tuple(rgb.shape) == (1, 3, H, W) and tuple(depth.shape) == (1, 1, H, W)
    rays = glue.gen_rays(E, Kt, W, H, torch.tensor([near], device=dev), torch.tensor([far], device=dev)).view(1, H * W, 8)
    r.seed, r._calls = 3, 0
    with torch.no_grad():
        ref = r(m, rays).fine
    assert torch.equal(rgb, ref.rgb.view(1, H, W, 3).permute(0, 3, 1, 2))
    assert torch.equal(depth, ref.depth.view(1, H, W, 1).permute(0, 3, 1, 2))
    assert float(rgb.min()) >= -1e-5 and float(rgb.max()) <= 1 + 1e-5
