"""Generate the golden vectors under ``tests/golden`` by running the UNMODIFIED reference render
path on the CPU (``oracle/ref_harness.py``).  Runs only in the build container, where
``/root/reference`` exists; the GPU box sees only the committed ``.npz`` files.

    python oracle/gen_golden.py            # (re)writes tests/golden/*.npz

Each fixture stores: the case configuration (json), the rays, every stage output of the
reference, and sha256 digests of the seeded inputs (scene maps, latent, MLP weights, noise)
so a drift of the generators in ``synthetic/synth.py`` is detected instead of silently
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

from synthetic import synth  # noqa: E402

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
    # the headline renderer parameters of BASELINE.json (K=128, G=48, NC=1000, 4 views, Facescape near/far, padding 32)
    "g4_nv4_k128_headline": dict(scene=dict(H=64, W=64, NV=4, seed=12, dataset="facescape", feature_padding=32),
                                 K=128, NC=1000, G=48, ray_stride=61, focal_scale=1.0, wseed=7, bias_scale=0.1, nseed=14),
    # BASELINE.json's stress configuration: 8 source views, 256 samples per ray (G = 96)
    "g5_nv8_k256_stress": dict(scene=dict(H=48, W=48, NV=8, seed=15, dataset="facescape", feature_padding=32),
                               K=256, NC=1000, G=96, ray_stride=53, focal_scale=1.0, wseed=16, bias_scale=0.1, nseed=17),
    # BASELINE.json configs[1]/[3] (cfg2 / cfg4): the DTU workload at the headline renderer parameters -- DTU near/far
    # (src/data/dtu.py:42-43), black background (configs/train_dtu.yaml:58), 4 views, K=128, G=48, NC=1000, NON-SQUARE maps
    # (64 x 80, the 512 x 640 aspect), dataset-faithful background sigma conf2std(0)
    "g6_nv4_k128_dtu_nonsquare": dict(scene=dict(H=64, W=80, NV=4, seed=18, dataset="dtu", feature_padding=32),
                                      K=128, NC=1000, G=48, ray_stride=71, focal_scale=1.0, wseed=19, bias_scale=0.1, nseed=20),
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


# ---- upstream wire format + camera-sweep poses -----------------------------------------------------
WIRE = dict(dtu=dict(seed=31, H=512, W=640, scale_factor=0.7 / 872.0), facescape=dict(seed=32, H=96, W=64))


def wire_inputs():
    """uint16 planes as TransMVSNet writes them (rebuilt from seeds by the tests): smooth surfaces with holes (zeros)."""
    out = {}
    c = WIRE["dtu"]
    rs = np.random.RandomState(c["seed"])
    yy, xx = np.meshgrid(np.arange(c["H"]), np.arange(c["W"]), indexing="ij")
    dep = (6000 + 2500 * np.sin(xx / 37.0) * np.cos(yy / 53.0) + rs.randint(0, 40, xx.shape)).astype(np.uint16)
    dep[rs.rand(*dep.shape) < 0.2] = 0
    conf = rs.randint(0, 10001, xx.shape).astype(np.uint16)      # confidence in [0,1] x 1e4
    out["dtu"] = dict(depth=dep, conf=conf)
    c = WIRE["facescape"]
    rs = np.random.RandomState(c["seed"])
    H, W = c["H"], c["W"]
    gt = rs.randint(0, 30000, (H, W)).astype(np.uint16)
    pred = rs.randint(8000, 26000, (H, W)).astype(np.uint16)
    cf = rs.randint(0, 10001, (H, W)).astype(np.uint16)
    mesh = rs.randint(8000, 26000, (H, W)).astype(np.uint16)
    for a in (pred, cf, mesh):
        a[rs.rand(H, W) < 0.35] = 0
    out["facescape"] = dict(gt=gt, pred=pred, conf=cf, mesh=mesh)
    return out


def pose_inputs():
    """49 DTU-like camera extrinsics on a dome looking at the origin (world->cam, OpenCV) and 2-4 Facescape source cameras."""
    rs = np.random.RandomState(41)
    ext = []
    for i in range(49):
        yaw, pitch = -0.9 + 1.8 * (i % 7) / 6.0, -0.5 + 0.7 * (i // 7) / 6.0
        e = synth.look_at_origin_w2c(yaw, 1.3 + 0.05 * rs.rand()).astype(np.float64)
        cp, sp = np.cos(pitch), np.sin(pitch)
        rx = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1.0]])
        ext.append((e @ rx).astype(np.float32))
    src = np.stack([synth.look_at_origin_w2c(y, 1.75) for y in (-0.45, 0.1, 0.5)]).astype(np.float32)
    # facescape's sweep assumes z-up world (y_ax = (0,0,-1)): rotate the rig so that its up axis is z
    to_zup = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1.0]], np.float32)
    src = np.stack([e @ to_zup for e in src])
    return dict(dtu_extrinsics=np.stack(ext), facescape_src_extrinsics=src)


def gen_wire(out_dir):
    """Outputs of the reference's own depth readers / conf2std / get_cam_sweep_extrinsics on PNGs written here with PIL.
    Full-size outputs are stored as sha256 digests + a strided sample (the DTU reader insists on 512 x 640)."""
    import tempfile
    from types import SimpleNamespace as NS
    import torch
    from PIL import Image
    from oracle import ref_harness as rh
    rh.install_stubs()
    from src.data.dtu import DTUDataSet
    from src.data.facescape import FacescapeDataSet
    w = wire_inputs()
    fx = {}

    def put(name, arr):
        arr = np.ascontiguousarray(arr)
        fx[name + "/sha256"] = digest(arr)
        fx[name + "/shape"] = np.array(arr.shape)
        fx[name + "/sample"] = arr.reshape(-1)[::97].copy()

    with tempfile.TemporaryDirectory() as td:
        # ---- DTU: read_depth on the prediction PNG and on the confidence PNG (dtu.py:100-119, 218-223), conf2std :68-70
        for ds in (1.0, 0.5):
            ns = NS(downsample=ds, scale_factor=WIRE["dtu"]["scale_factor"])
            for key in ("depth", "conf"):
                Image.fromarray(w["dtu"][key]).save(f"{td}/{key}.png")
            d, m = DTUDataSet.read_depth(ns, f"{td}/depth.png")
            c, _ = DTUDataSet.read_depth(ns, f"{td}/conf.png")
            std = DTUDataSet._getconf2std(ns)(c)
            put(f"dtu/ds{ds}/depth", d.numpy()), put(f"dtu/ds{ds}/mask", m.numpy()), put(f"dtu/ds{ds}/std", std.numpy())
        # ---- Facescape: one PNG of three panels gt | pred | conf + the mesh depth PNG (facescape.py:80-104), conf2std :54-56
        f = w["facescape"]
        Image.fromarray(np.concatenate([f["gt"], f["pred"], f["conf"]], axis=1)).save(f"{td}/d3.png")
        Image.fromarray(f["mesh"]).save(f"{td}/mesh.png")
        conf2std = FacescapeDataSet._getconf2std(NS())
        for dt in ("original", "mesh", "merge"):
            p_, c_ = FacescapeDataSet.read_depth(f"{td}/d3.png", f"{td}/mesh.png", depth_type=dt)
            put(f"facescape/{dt}/depth", p_.numpy()), put(f"facescape/{dt}/std", conf2std(c_).numpy())
    # ---- camera sweeps (dtu.py:246-340, facescape.py:365-423)
    pi = pose_inputs()
    ns = NS(cam_dict=dict(extrinsics=torch.from_numpy(pi["dtu_extrinsics"])))
    for nf in (5, 12):
        fx[f"poses/dtu/{nf}"] = DTUDataSet.get_cam_sweep_extrinsics(ns, nf).numpy()
    src = torch.from_numpy(pi["facescape_src_extrinsics"])
    ns = NS(range_hor=45, __getitem__=None)
    ns.__getitem__ = lambda idx: dict(target_extrinsics=src[0], src_extrinsics=src)
    for nf, kw in ((7, {}), (4, dict(radius=1.5, sweep_range=30))):
        fx[f"poses/facescape/{nf}"] = FacescapeDataSet.get_cam_sweep_extrinsics(ns, nf, 0, **kw).numpy()
    fx["digests"] = json.dumps(dict(inputs=digest(*[w["dtu"][k] for k in ("depth", "conf")], *[w["facescape"][k] for k in ("gt", "pred", "conf", "mesh")],
                                                  pi["dtu_extrinsics"], pi["facescape_src_extrinsics"])))
    np.savez_compressed(out_dir / "wire.npz", **fx)
    print(f"wire: {len(fx)} entries, dtu depth sample {fx['dtu/ds1.0/depth/sample'][:3]}, facescape sweep {fx['poses/facescape/7'].shape}")


TRAIN_CASE = dict(scene=dict(H=16, W=16, NV=2, seed=12, dataset="facescape", feature_padding=4), K=8, NC=64, G=3,
                  ray_stride=8, focal_scale=1.0, wseed=13, bias_scale=0.1, nseed=14, cseed=15)


# second training case: DTU near/far and black background (white_bkgd False), 3 views, and a cotangent on the
# compositing weights as well (the reference's losses may use them, src/models/diner.py:262-290)
TRAIN_CASE_DTU = dict(scene=dict(H=20, W=16, NV=3, seed=21, dataset="dtu", feature_padding=4), K=12, NC=96, G=4,
                      ray_stride=9, focal_scale=1.0, wseed=7, bias_scale=0.1, nseed=23, cseed=24, weights_cotangent=True)
TRAIN_CASES = {"train": TRAIN_CASE, "train_dtu": TRAIN_CASE_DTU}


def weights_cotangent(NR, K, cseed):
    return np.random.RandomState(cseed + 1000).standard_normal((1, NR, K)).astype(np.float32)


def train_cotangents(NR, cseed):
    rs = np.random.RandomState(cseed)
    return rs.standard_normal((1, NR, 3)).astype(np.float32), rs.standard_normal((1, NR)).astype(np.float32)


def grad_probe_indices(shape, n=64, seed=99):
    rs = np.random.RandomState(seed + int(np.prod(shape)) % 1000)
    return rs.randint(0, int(np.prod(shape)), size=min(n, int(np.prod(shape))))


def gen_train(out_dir, name="train"):
    """Gradients of L = <c_rgb, rgb> + <c_depth, depth> (+ <c_w, weights>) w.r.t. every MLP parameter and the latent,
    from the reference's own autograd (composite + PixelNeRF.forward + ResnetFC on the reference's sorted samples)."""
    import torch
    from oracle import ref_harness as rh
    cfg = TRAIN_CASES[name]
    sc, w, rays, noise = case_inputs(cfg)
    nerf = rh.build_model(sc, w)
    ref = rh.run_reference(nerf, rays, cfg["K"], cfg["NC"], cfg["G"], noise, white_bkgd=sc.white_bkgd, want_internals=False)
    z = torch.from_numpy(ref["z_fill"])
    nerf.encoder.latent = nerf.encoder.latent.clone().requires_grad_(True)
    for p in nerf.mlp_fine.parameters():
        p.requires_grad_(True)
    rend = rh.import_reference().NeRFRendererDGS(n_samples=cfg["K"], n_depth_candidates=cfg["NC"], n_gaussian=cfg["G"],
                                                 white_bkgd=sc.white_bkgd)
    weights, rgb, depth = rend.composite(nerf, torch.from_numpy(rays), z)
    c_rgb, c_depth = train_cotangents(rays.shape[1], cfg["cseed"])
    loss = (rgb * torch.from_numpy(c_rgb)).sum() + (depth * torch.from_numpy(c_depth)).sum()
    if cfg.get("weights_cotangent"):
        loss = loss + (weights * torch.from_numpy(weights_cotangent(rays.shape[1], cfg["K"], cfg["cseed"]))).sum()
    loss.backward()
    fixture = dict(config=json.dumps(cfg), z_fill=ref["z_fill"], rgb=rgb.detach().numpy(), depth=depth.detach().numpy(),
                   latent_grad=nerf.encoder.latent.grad.numpy())
    for pname, p in nerf.mlp_fine.named_parameters():
        gnp = p.grad.numpy()
        idx = grad_probe_indices(gnp.shape)
        fixture[f"g_sum/{pname}"] = np.float64(gnp.astype(np.float64).sum())
        fixture[f"g_norm/{pname}"] = np.float64(np.sqrt((gnp.astype(np.float64) ** 2).sum()))
        fixture[f"g_probe/{pname}"] = gnp.reshape(-1)[idx]
    np.savez_compressed(out_dir / f"{name}.npz", **fixture)
    print(f"{name}: NR={rays.shape[1]} |latent_grad|={np.abs(fixture['latent_grad']).max():.3e} "
          f"|g lin_out.w|={fixture['g_norm/lin_out.weight']:.3e} |g lin_in.w|={fixture['g_norm/lin_in.weight']:.3e}")


def main():
    from oracle import ref_harness as rh
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    if "--wire-only" in sys.argv:
        gen_wire(out_dir)
        return
    if not any(a.startswith("--case=") for a in sys.argv):
        gen_wire(out_dir)
        gen_glue(out_dir)
        if "--glue-only" in sys.argv:
            return
        if "--train-dtu-only" in sys.argv:   # adds the second training case without touching the first
            gen_train(out_dir, "train_dtu")
            return
        for tname in TRAIN_CASES:
            gen_train(out_dir, tname)
        if "--train-only" in sys.argv:
            return
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--case=")]   # regenerate selected cases only
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
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
