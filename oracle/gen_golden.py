"""Generate the golden vectors under ``tests/golden`` by running the UNMODIFIED reference render
path on the CPU (``oracle/ref_harness.py``).  Runs only in the build container, where
``/root/reference`` exists; the GPU box sees only the committed ``.npz`` files.

    python oracle/gen_golden.py            # (re)writes tests/golden/*.npz

Each fixture stores: the case configuration (json), the rays, every stage output of the
reference, and sha256 digests of the seeded inputs (scene maps, latent, MLP weights, noise)
so a drift of the generators in ``diner_amd/synth.py`` is detected instead of silently
comparing against different inputs.  Inputs themselves are rebuilt from the seeds.
"""
from __future__ import annotations

import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from diner_amd import synth  # noqa: E402

CASES = {
    # name: scene kwargs, renderer config, ray selection
    "g0_nv4_k16": dict(scene=dict(H=32, W=32, NV=4, seed=0, dataset="facescape", feature_padding=4),
                       K=16, NC=200, G=6, ray_stride=4, focal_scale=1.0, wseed=1, bias_scale=0.0, nseed=2),
    "g1_nv2_k64_dtu": dict(scene=dict(H=48, W=40, NV=2, seed=3, dataset="dtu", feature_padding=4,
                                      bg_sigma_zero=True),
                           K=64, NC=1000, G=24, ray_stride=15, focal_scale=1.0, wseed=4, bias_scale=0.1, nseed=5),
    "g2_nv4_k64_pad32": dict(scene=dict(H=64, W=64, NV=4, seed=6, dataset="facescape", feature_padding=32),
                             K=64, NC=1000, G=24, ray_stride=32, focal_scale=1.0, wseed=7, bias_scale=0.1, nseed=8),
    "g3_nv3_k40_wide": dict(scene=dict(H=32, W=32, NV=3, seed=9, dataset="facescape", feature_padding=4),
                            K=40, NC=1000, G=15, ray_stride=8, focal_scale=0.45, wseed=10, bias_scale=0.1, nseed=11),
}
N_FULL_INPUT_POINTS = 16   # points whose full 567-vector is stored
N_TAIL_INPUT_POINTS = 512  # points whose 55 non-latent inputs are stored


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def case_inputs(cfg):
    """Rebuild every input of a case from its seeds (shared by the generator and the tests)."""
    sc = synth.make_scene(**cfg["scene"])
    w = synth.make_mlp_weights(cfg["wseed"], bias_scale=cfg["bias_scale"])
    rays = sc.target_rays(focal_scale=cfg["focal_scale"])[:, ::cfg["ray_stride"]]
    noise = synth.make_noise(rays.shape[1], cfg["NC"], cfg["G"], cfg["K"], seed=cfg["nseed"])
    return sc, w, np.ascontiguousarray(rays), noise


def input_digests(sc, w, rays, noise):
    return dict(maps=digest(sc.poses, sc.focal, sc.c, sc.depths, sc.depths_std, sc.normals),
                latent=digest(sc.latent), weights=digest(*[w[k] for k in sorted(w)]),
                rays=digest(rays), noise=digest(*noise))


def glue_inputs():
    """Inputs of the adjacent per-image producers (gen_rays, depth2normal): two cameras, depth maps with
    background, holes and a pixel on the image border."""
    rs = np.random.RandomState(77)
    H, W = 20, 24
    ext = np.stack([synth.look_at_origin_w2c(0.3, 1.75), synth.look_at_origin_w2c(-0.45, 1.6)])
    k = np.stack([synth.intrinsics(W, H), synth.intrinsics(W, H)])
    k[1, 0, 0] *= 0.9
    k[1, 0, 2] += 1.25
    k[1, 1, 2] -= 0.75
    near, far = np.array([1.0, 0.321], np.float32), np.array([2.5, 1.204], np.float32)
    depth = np.stack([synth.sphere_zdepth(ext[i], k[i], W, H, 0.45) for i in range(2)])[:, None]
    depth = (depth * (1 + 0.02 * rs.standard_normal(depth.shape)) * (depth > 0)).astype(np.float32)
    depth[0, 0, 9:11, 11:13] = 0      # a hole inside the object
    depth[1, 0, 0, :] = 1.3           # a valid row on the image border
    return dict(extrinsics=ext.astype(np.float32), intrinsics=k.astype(np.float32), z_near=near, z_far=far, H=H, W=W,
                dmap=depth)


def gen_glue(out_dir):
    import torch
    from oracle import ref_harness as rh
    rh.import_reference()
    from src.util.cam_geometry import gen_rays
    from src.util.depth2normal import depth2normal
    g = glue_inputs()
    t = torch.from_numpy
    rays = gen_rays(t(g["extrinsics"]), t(g["intrinsics"]), g["W"], g["H"], t(g["z_near"]), t(g["z_far"])).numpy()
    normals = depth2normal(t(g["dmap"]), t(g["intrinsics"])).numpy()
    np.savez_compressed(out_dir / "glue.npz", digests=json.dumps(dict(inputs=digest(*[g[k] for k in ("extrinsics", "intrinsics", "z_near", "z_far", "dmap")]))),
                        rays=rays, normals=normals)
    print(f"glue: rays {rays.shape} normals {normals.shape} nan={np.isnan(normals).sum()}")


def main():
    from oracle import ref_harness as rh
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    gen_glue(out_dir)
    if "--glue-only" in sys.argv:
        return
    for name, cfg in CASES.items():
        t0 = time.time()
        sc, w, rays, noise = case_inputs(cfg)
        nerf = rh.build_model(sc, w)
        ref = rh.run_reference(nerf, rays, cfg["K"], cfg["NC"], cfg["G"], noise, white_bkgd=sc.white_bkgd)
        NV, P = ref["mlp_input"].shape[1:3]
        sel_full = np.linspace(0, P - 1, N_FULL_INPUT_POINTS).astype(np.int64)
        sel_tail = np.linspace(0, P - 1, min(P, N_TAIL_INPUT_POINTS)).astype(np.int64)
        fixture = dict(
            config=json.dumps(cfg), digests=json.dumps(input_digests(sc, w, rays, noise)),
            rays=rays, hit=ref["hit"], n_missing=ref["n_missing"].astype(np.int32),
            z_cand=ref["z_cand"][0], pt_likelihood=ref["pt_likelihood"][0],
            z_dg=ref["z_dg"][0], z_fill=ref["z_fill"][0], rgbsigma=ref["rgbsigma"][0],
            weights=ref["weights"][0], rgb=ref["rgb"][0], depth=ref["depth"][0],
            mlp_input_sel_full=sel_full, mlp_input_full=ref["mlp_input"][0][:, sel_full],
            mlp_input_sel_tail=sel_tail, mlp_input_tail=ref["mlp_input"][0][:, sel_tail, 512:],
        )
        path = out_dir / f"{name}.npz"
        np.savez_compressed(path, **fixture)
        print(f"{name}: NR={rays.shape[1]} hit={ref['hit'].mean():.2f} missing/ray={ref['n_missing'].mean():.1f} "
              f"sigma_max={ref['rgbsigma'][..., 3].max():.2f} -> {path.name} {path.stat().st_size/1e6:.2f} MB "
              f"({time.time()-t0:.1f}s)")


if __name__ == "__main__":
    main()
