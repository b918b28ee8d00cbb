"""Camera-sweep pose generators against the outputs of the reference's own
``get_cam_sweep_extrinsics`` methods (tests/golden/wire.npz, oracle/gen_golden.py --wire-only).
Tolerance: the reference works in float32 (linalg.inv, lstsq, acos near the arc ends) -> 2e-5 absolute on matrix entries."""
import json

import numpy as np

from diner_amd import poses
from oracle.gen_golden import digest, pose_inputs, wire_inputs
from tests.conftest import GOLDEN_DIR

G = dict(np.load(GOLDEN_DIR / "wire.npz", allow_pickle=False))


def test_inputs_match_the_fixture():
    w, pi = wire_inputs(), pose_inputs()
    got = digest(*[w["dtu"][k] for k in ("depth", "conf")], *[w["facescape"][k] for k in ("gt", "pred", "conf", "mesh")],
                 pi["dtu_extrinsics"], pi["facescape_src_extrinsics"])
    assert got == json.loads(str(G["digests"]))["inputs"]


def test_dtu_cam_sweep():
    pi = pose_inputs()
    for nf in (5, 12):
        ref = G[f"poses/dtu/{nf}"]
        got = poses.dtu_cam_sweep_extrinsics(pi["dtu_extrinsics"], nf)
        assert got.shape == ref.shape == (nf, 4, 4) and got.dtype == np.float32
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    # the sweep starts with the left camera's rotation, passes the centre one's and ends with the right one's (the centres are
    # re-projected onto a common radius, so only the rotations are reproduced)
    np.testing.assert_allclose(poses.dtu_cam_sweep_extrinsics(pi["dtu_extrinsics"], 5)[[0, 2, 4], :3, :3], pi["dtu_extrinsics"][[11, 24, 18], :3, :3], atol=2e-5)
    r = got[:, :3, :3]
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", r, r), np.broadcast_to(np.eye(3), r.shape), atol=1e-5)


def test_facescape_cam_sweep():
    pi = pose_inputs()
    for nf, kw in ((7, {}), (4, dict(radius=1.5, sweep_range=30))):
        ref = G[f"poses/facescape/{nf}"]
        got = poses.facescape_cam_sweep_extrinsics(pi["facescape_src_extrinsics"], nf, **kw)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    c = -np.einsum("nji,nj->ni", got[:, :3, :3], got[:, :3, 3])
    np.testing.assert_allclose(np.linalg.norm(c, axis=-1), 1.5, atol=1e-5)          # all on the sweep circle
    np.testing.assert_allclose(c[:, 2], c[0, 2], atol=1e-5)                           # around the world z axis
