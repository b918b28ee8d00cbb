"""The adjacent per-image producers: gen_rays and depth2normal.
CPU: the numpy restatements in diner_amd/synth.py against goldens from the unmodified reference.
GPU: the HIP kernels (C ABI) against the same goldens."""
import json

import numpy as np
import pytest

from diner_amd import synth
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
