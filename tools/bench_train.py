#!/usr/bin/env python3
"""Training-step timing of the differentiable path: forward with saved activations +
backward for one reference-sized training batch (4096 rays = the 64x64 patch of DINER.calc_losses,
src/models/diner.py:232-261; K=40, n_gaussian=15, NC=1000 = configs/train_diner_facescape.yaml:61-66).
Prints one JSON line.  Not the headline metric (bench.py is)."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diner_amd import NeRFRendererDGS  # noqa: E402
from synthetic import synth  # noqa: E402
from synthetic.model_stub import model_from_scene  # noqa: E402


def main(NV=4, H=256, W=256, NR=4096, K=40, G=15, NC=1000, steps=3):
    dev = torch.device("cuda:0")
    sc = synth.make_scene(H, W, NV, seed=0, with_latent=False)
    h, w = sc.latent_hw
    latent = torch.randn((1, NV, 512, h, w), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    m = model_from_scene(sc, synth.make_mlp_weights(1, bias_scale=0.1), device=dev, latent=latent)
    for p in m.mlp_fine.parameters():
        p.requires_grad_(True)
    m.encoder.latent.requires_grad_(True)
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G)
    rays = torch.from_numpy(sc.target_rays(crop=(H // 2 - 32, W // 2 - 32, 64, 64))).to(dev)
    assert rays.shape[1] == NR
    tgt = torch.rand((1, NR, 3), device=dev)
    times = []
    for i in range(steps + 1):
        for p in m.mlp_fine.parameters():
            p.grad = None
        m.encoder.latent.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = r(m, rays)
        loss = ((out.fine.rgb - tgt) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = min(times[1:])
    flop = 3 * NR * K * 2 * (NV * 2_387_456 + 1_050_624)  # forward + 2x backward
    print(json.dumps({"what": "training step (sampler + forward + backward), 4096 rays x 40 samples x %d views" % NV,
                      "ms_per_step": t * 1e3, "rays_per_s": NR / t, "tflops_fwd_bwd": flop / t / 1e12,
                      "mlp_grad_norm": float(sum(float(p.grad.norm()) ** 2 for p in m.mlp_fine.parameters()) ** 0.5),
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
