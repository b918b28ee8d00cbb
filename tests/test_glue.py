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


@pytest.mark.gpu
def test_render_image_abi_rays_out_and_two_scenes():
    """``diner_render_image`` through the C-ABI: the rays the sampler generates (``rays_out``) are bit-identical to
    ``diner_gen_rays`` (src/util/cam_geometry.py:36-79) for two scenes with different target cameras, and the image equals
    ``diner_render`` on those rays with the same seed."""
    import ctypes as C
    import torch
    from diner_amd import NeRFRendererDGS, _lib, glue
    from diner_amd.renderer import _ptr, _stream, check
    from synthetic import synth
    from synthetic.model_stub import model_from_scene
    dev = torch.device("cuda:0")
    import copy
    a, b = synth.make_scene(24, 32, 2, seed=9, feature_padding=4), synth.make_scene(24, 32, 2, seed=10, feature_padding=4)
    sc = copy.copy(a)
    for name in ("poses", "focal", "c", "depths", "depths_std", "normals", "latent"):
        setattr(sc, name, np.concatenate([getattr(a, name), getattr(b, name)], 0))
    m = model_from_scene(sc, synth.make_mlp_weights(2, bias_scale=0.1), device=dev)
    r = NeRFRendererDGS(n_samples=16, n_depth_candidates=128, n_gaussian=5, white_bkgd=sc.white_bkgd)
    H, W, K = 12, 20, 16
    E0 = torch.from_numpy(np.ascontiguousarray(sc.target_extrinsics, dtype=np.float32)).to(dev)
    E = torch.stack([E0, E0.clone()])
    E[1, :3, 3] += torch.tensor([0.02, -0.01, 0.03], device=dev)
    Kt = torch.tensor([[[1.2 * W, 0, W / 2], [0, 1.2 * W, H / 2], [0, 0, 1]], [[1.1 * W, 0, W / 2 + 1], [0, 1.3 * W, H / 2 - 1], [0, 0, 1]]],
                      dtype=torch.float32, device=dev)
    zn = torch.tensor([float(sc.near), float(sc.near) * 1.05], device=dev)
    zf = torch.tensor([float(sc.far), float(sc.far) * 0.95], device=dev)
    packed = r._mlp(m)
    scn, _keep = r._scene(m, need_latent=True, packed_mlp=packed)
    cfg = r._cfg(K, 128, 5)
    cam = _lib.DinerTargetCam()
    cam.extrinsics, cam.intrinsics, cam.z_near, cam.z_far, cam.H, cam.W = E.data_ptr(), Kt.data_ptr(), zn.data_ptr(), zf.data_ptr(), H, W
    L, prec = _lib.lib(), _lib.PRECISIONS[r.precision]
    ws = torch.empty(int(L.diner_render_image_workspace_floats(2, H, W, K, scn.NV, prec)), dtype=torch.float32, device=dev)
    rays_out = torch.full((2, H * W, 8), float("nan"), device=dev)
    rgb, depth = torch.empty((2, H * W, 3), device=dev), torch.empty((2, H * W), device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.diner_render_image(C.byref(scn), _ptr(packed), C.byref(cam), C.byref(cfg), int(bool(sc.white_bkgd)), prec, 77, _ptr(ws), _ptr(rays_out),
                               _ptr(rgb), _ptr(depth), None, _ptr(status), _stream(dev)), "diner_render_image")
    rays = glue.gen_rays(E, Kt, W, H, zn, zf).view(2, H * W, 8)
    assert torch.equal(rays_out, rays)
    ws2 = torch.empty(int(L.diner_render_workspace_floats(2, H * W, K, scn.NV, prec)), dtype=torch.float32, device=dev)
    rgb2, depth2 = torch.empty_like(rgb), torch.empty_like(depth)
    check(L.diner_render(C.byref(scn), _ptr(packed), _ptr(rays), H * W, C.byref(cfg), int(bool(sc.white_bkgd)), prec, None, None, None, 77, _ptr(ws2),
                         _ptr(rgb2), _ptr(depth2), None, _ptr(status), _stream(dev)), "diner_render")
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2) and int(status.cpu()[0]) == 0
    # NULL camera / bad size are refused, not launched
    assert L.diner_render_image(C.byref(scn), _ptr(packed), None, C.byref(cfg), 0, prec, 0, _ptr(ws), None, _ptr(rgb), _ptr(depth), None, None, None) != 0
    cam.H = 0
    assert L.diner_render_image(C.byref(scn), _ptr(packed), C.byref(cam), C.byref(cfg), 0, prec, 0, _ptr(ws), None, _ptr(rgb), _ptr(depth), None, None, None) != 0
