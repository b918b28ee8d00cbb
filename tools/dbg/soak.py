"""Soak of the product path: the same frame rendered N times per configuration, every output compared BIT FOR BIT with the first one
(the kernels have no atomics and no order-dependent reductions on this path: any difference is a hazard or a race).  Configurations: views
in tile (NV = 4, 8), view-sequential (NV = 3), lin_z as maps and as per-point GEMMs.   usage: soak.py [seconds per configuration]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
for H, NV, K, G, prec in ((256, 4, 128, 48, "f16x3"), (256, 8, 64, 24, "f16x3"), (256, 3, 64, 24, "f16x3"), (128, 4, 64, 24, "f16x3-gemm"),
                          (128, 2, 64, 24, "f16x3-gemm"), (128, 3, 40, 15, "f16x3-gemm")):
    sc = synth.make_scene(H, H, NV, seed=0, with_latent=False)
    h, w = sc.latent_hw
    latent = torch.randn((1, NV, 512, h, w), generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
    m = model_from_scene(sc, synth.make_mlp_weights(7, bias_scale=0.1), device=dev, latent=latent)
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=1000, n_gaussian=G)
    r.precision = "f16x3"
    if prec == "f16x3-gemm": r.linz_maps = False
    rays = torch.from_numpy(sc.target_rays()).to(dev)
    with torch.no_grad():
        # one set of samples (the sampler's in-kernel RNG advances from call to call), then the same frame over and over
        z = r.fill_up_uniform_samples(r.sample_depthguided(rays, m, K, 1000), rays)
        first, n, bad, t0 = None, 0, 0, time.time()
        while time.time() - t0 < budget:
            out = r(m, rays, want_weights=True, z_samples=z)
            cur = torch.cat([out.fine.rgb.reshape(-1), out.fine.depth.reshape(-1), out.fine.weights.reshape(-1)])
            if first is None: first = cur.clone()
            elif not torch.equal(cur, first):
                bad += 1
                d = (cur - first).abs()
                if bad <= 5: print("   frame %d differs: %d values, max %.3e" % (n, int((d > 0).sum()), float(d.max())), flush=True)
            n += 1
    print("%dx%d NV=%d K=%d %-11s %5d frames, %d differ from the first; finite: %s" % (H, H, NV, K, prec, n, bad, bool(torch.isfinite(first).all())), flush=True)
