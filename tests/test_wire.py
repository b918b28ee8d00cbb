"""Upstream wire format: uint16 depth / confidence planes -> depths, depths_std.
Pinned by tests/golden/wire.npz = outputs of the reference's own ``DTUDataSet.read_depth`` / ``FacescapeDataSet.read_depth`` /
``conf2std`` on uint16 PNGs (oracle/gen_golden.py --wire-only): full arrays as sha256 digests + a strided sample.
CPU: the numpy restatement bit-exact against that fixture (and hand-computed values).  GPU: the HIP decoder bit-exact against
the fixture and the restatement, feeding the packed maps of the renderer."""
import hashlib

import numpy as np
import pytest

from oracle import wire_oracle as wo
from oracle.gen_golden import WIRE, wire_inputs
from tests.conftest import GOLDEN_DIR

F = np.float32
G = dict(np.load(GOLDEN_DIR / "wire.npz", allow_pickle=False))


def same_as_fixture(name, arr):
    """bit-exact: shape, sha256 of the bytes, and the stored strided sample"""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    assert tuple(G[name + "/shape"]) == arr.shape, (name, arr.shape)
    np.testing.assert_array_equal(arr.reshape(-1)[::97], G[name + "/sample"], err_msg=name)
    assert hashlib.sha256(arr.tobytes()).hexdigest() == str(G[name + "/sha256"]), name


def test_restatement_matches_the_reference_readers():
    w = wire_inputs()
    sf = WIRE["dtu"]["scale_factor"]
    for ds, stride in ((1.0, 1), (0.5, 2)):
        d, m = wo.dtu_read_depth(w["dtu"]["depth"][None], sf, stride)
        c, _ = wo.dtu_read_depth(w["dtu"]["conf"][None], sf, stride)
        same_as_fixture(f"dtu/ds{ds}/depth", d), same_as_fixture(f"dtu/ds{ds}/mask", m), same_as_fixture(f"dtu/ds{ds}/std", wo.dtu_conf2std(c))
    f = w["facescape"]
    for dt in ("original", "mesh", "merge"):
        p, c = wo.facescape_read_depth(f["pred"][None], f["conf"][None], None if dt == "original" else f["mesh"][None], depth_type=dt)
        same_as_fixture(f"facescape/{dt}/depth", p), same_as_fixture(f"facescape/{dt}/std", wo.facescape_conf2std(c))


@pytest.mark.gpu
def test_hip_decoder_matches_the_reference_readers():
    from diner_amd import wire
    w = wire_inputs()
    sf = WIRE["dtu"]["scale_factor"]
    for ds in (1.0, 0.5):
        d, s, m = wire.decode_dtu(w["dtu"]["depth"][None], w["dtu"]["conf"][None], sf, downsample=ds, want_mask=True)
        same_as_fixture(f"dtu/ds{ds}/depth", d.cpu().numpy()[0]), same_as_fixture(f"dtu/ds{ds}/mask", m.cpu().numpy()[0])
        same_as_fixture(f"dtu/ds{ds}/std", s.cpu().numpy()[0])
    f = w["facescape"]
    zero = np.zeros_like(f["pred"])
    for dt, (dep, con, mesh) in dict(original=(f["pred"], f["conf"], None), merge=(f["pred"], f["conf"], f["mesh"]),
                                     mesh=(zero, zero, f["mesh"])).items():   # 'mesh' = merge with empty MVS planes
        d, s = wire.decode_facescape(dep[None], con[None], None if mesh is None else mesh[None])
        same_as_fixture(f"facescape/{dt}/depth", d.cpu().numpy()[0]), same_as_fixture(f"facescape/{dt}/std", s.cpu().numpy()[0])


def test_restatement_known_answers():
    u = np.array([[[0, 1, 10000, 65535]]], dtype=np.uint16)
    d, m = wo.dtu_read_depth(u, 0.7 / 872.0)
    # scale_factor == the training scale: the division and the multiplication cancel up to rounding
    np.testing.assert_allclose(d[0, 0], [0.0, 1e-4, 1.0, 6.5535], rtol=3e-7)
    np.testing.assert_array_equal(m[0, 0], [0, 1, 1, 1])
    assert d.dtype == np.float32
    s = wo.dtu_conf2std(F(0.5))
    assert s == F(F(-2.5679e-2) * F(0.5)) + F(3.2818e-2) and abs(float(s) - 0.0199785) < 1e-7
    # background (confidence 0) -> the dataset-faithful non-zero sigma quoted in SURVEY.md §8(d)
    assert abs(float(wo.dtu_conf2std(F(0))) - 0.032818) < 1e-8 and abs(float(wo.facescape_conf2std(F(0))) - 0.01649) < 1e-8
    # Facescape merge: the mesh depth wins where it is non-zero, the MVS prediction fills its holes
    dep = np.array([[[20000, 0, 15000, 0]]], dtype=np.uint16)
    con = np.array([[[9000, 0, 5000, 0]]], dtype=np.uint16)
    mesh = np.array([[[0, 0, 14000, 13000]]], dtype=np.uint16)
    p, c = wo.facescape_read_depth(dep, con, mesh)
    np.testing.assert_array_equal(p[0, 0], (np.array([20000, 0, 14000, 13000], F) * F(1e-4)))
    np.testing.assert_array_equal(c[0, 0], np.array([F(9000) * F(1e-4), 0, F(0.8), F(0.8)], F))
    p0, c0 = wo.facescape_read_depth(dep, con)
    np.testing.assert_array_equal(p0, dep.astype(F) * F(1e-4))


def test_nearest_downsample_is_a_stride():
    u = np.arange(8 * 12, dtype=np.uint16).reshape(1, 8, 12)
    d, _ = wo.dtu_read_depth(u, 1.0, stride=2)
    assert d.shape == (1, 4, 6)
    np.testing.assert_array_equal(d[0], (u[0, ::2, ::2].astype(F) * F(1e-4)) / F(0.7 / 872.0) * F(1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1, 2])
def test_hip_decoder_dtu_bit_exact(stride):
    from diner_amd import wire
    rs = np.random.RandomState(0)
    dep = rs.randint(0, 65536, size=(3, 64, 80)).astype(np.uint16)
    dep[rs.rand(*dep.shape) < 0.3] = 0
    con = rs.randint(0, 65536, size=(3, 64, 80)).astype(np.uint16)
    sf = 0.7 / 872.0
    d, s, m = wire.decode_dtu(dep, con, sf, downsample=1.0 / stride, want_mask=True)
    rd, rm = wo.dtu_read_depth(dep, sf, stride)
    rc, _ = wo.dtu_read_depth(con, sf, stride)
    np.testing.assert_array_equal(d.cpu().numpy()[:, 0], rd)
    np.testing.assert_array_equal(m.cpu().numpy()[:, 0], rm)
    np.testing.assert_array_equal(s.cpu().numpy()[:, 0], wo.dtu_conf2std(rc))


@pytest.mark.gpu
@pytest.mark.parametrize("merge", [False, True])
def test_hip_decoder_facescape_bit_exact(merge):
    from diner_amd import wire
    rs = np.random.RandomState(1)
    dep = rs.randint(0, 30000, size=(2, 48, 48)).astype(np.uint16)
    con = rs.randint(0, 10000, size=(2, 48, 48)).astype(np.uint16)
    mesh = rs.randint(0, 30000, size=(2, 48, 48)).astype(np.uint16)
    for a in (dep, con, mesh):
        a[rs.rand(*a.shape) < 0.4] = 0
    d, s = wire.decode_facescape(dep, con, mesh if merge else None)
    rp, rc = wo.facescape_read_depth(dep, con, mesh if merge else None)
    np.testing.assert_array_equal(d.cpu().numpy()[:, 0], rp)
    np.testing.assert_array_equal(s.cpu().numpy()[:, 0], wo.facescape_conf2std(rc))


@pytest.mark.gpu
def test_decoded_planes_feed_the_map_packing():
    """wire planes -> depths/std -> packed maps (depth2normal fused): the encode-side chain entirely on the GPU."""
    import torch
    from diner_amd import glue, wire
    from synthetic import synth
    sc = synth.make_scene(32, 32, 2, seed=3, feature_padding=4)
    dep = np.clip(np.round(sc.depths[0, :, 0] / 1e-4), 0, 65535).astype(np.uint16)        # encode the synthetic depth as a PNG would
    std = np.clip(np.round((sc.depths_std[0, :, 0] - 1.649e-2) / -1.582e-2 / 1e-4), 0, 65535).astype(np.uint16)
    d, s = wire.decode_facescape(dep, std)
    intr = np.zeros((1, 2, 3, 3), np.float32)
    intr[..., 0, 0] = sc.focal[..., 0]; intr[..., 1, 1] = sc.focal[..., 1]; intr[..., 0, 2] = sc.c[..., 0]; intr[..., 1, 2] = sc.c[..., 1]; intr[..., 2, 2] = 1
    maps = glue.pack_maps_from_depth(d[None], s[None], torch.from_numpy(intr).to(d.device))
    assert tuple(maps.shape) == (1, 2, 32, 32, 8)
    np.testing.assert_allclose(maps[0, :, :, :, 3].cpu().numpy(), sc.depths[0, :, 0], atol=6e-5)   # u16 quantisation: 1e-4 / 2
    assert bool(torch.isfinite(maps).all())
