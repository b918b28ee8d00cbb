"""Does the ORDER of the rays matter to the point/MLP kernel ("2-D ray blocks")?  One cfg3 frame rendered with its rays in
row-major pixel order (what gen_rays hands the renderer), in B x B pixel blocks (B = 8, 16, 32) and in a random order; kernel time by the
renderer's stage events, interleaved rounds, one process.  A tile of the kernel = 64 consecutive points of that order (half a ray at K = 128),
consecutive tiles run on the workgroups of one XCD."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
H = W = 512; NV, K, G, NC = 4, 128, 48, 1000
sc = synth.make_scene(H, W, NV, seed=0, with_latent=False)
h, w = sc.latent_hw
latent = torch.randn((1, NV, 512, h, w), generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
m = model_from_scene(sc, synth.make_mlp_weights(7, bias_scale=0.1), device=dev, latent=latent)
r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G)
rays = torch.from_numpy(sc.target_rays()).to(dev)          # [1, H*W, 8], row-major
idx = np.arange(H * W).reshape(H, W)
orders = {"row-major": idx.reshape(-1)}
for B in (8, 16, 32):
    orders["%dx%d blocks" % (B, B)] = idx.reshape(H // B, B, W // B, B).transpose(0, 2, 1, 3).reshape(-1)
orders["random"] = np.random.default_rng(0).permutation(H * W)
res = {k: [] for k in orders}
with torch.no_grad():
    for rnd in range(3):
        for name, o in orders.items():
            rr = rays[:, torch.from_numpy(o).to(dev)].contiguous()
            r.stage_events = []
            out = r(m, rr); torch.cuda.synchronize()
            e = r.stage_events[0]
            res[name].append(e[1].elapsed_time(e[2]))
for name, ts in res.items():
    print("%-14s point/MLP kernel min %.1f ms  %s" % (name, min(ts[1:]), ["%.1f" % t for t in ts]), flush=True)
