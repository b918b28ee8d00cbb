#!/usr/bin/env python3
"""Per-launch durations of the point/MLP kernel over consecutive frames (clock ramp / throttling check)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene

dev = torch.device("cuda:0")
sc = synth.make_scene(512, 512, 4, seed=0, dataset="facescape", with_latent=False)
h, w = sc.latent_hw
lat = torch.randn((1, 4, 512, h, w), device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
m = model_from_scene(sc, synth.make_mlp_weights(7, bias_scale=0.1), device=dev, latent=lat)
r = NeRFRendererDGS(n_samples=128, n_depth_candidates=1000, n_gaussian=48, white_bkgd=True)
rays = torch.from_numpy(sc.target_rays()).to(dev)
r.stage_events = []
with torch.no_grad():
    for _ in range(10):
        r(m, rays)
torch.cuda.synchronize()
print("mlp kernel ms per frame:", ["%.1f" % e[1].elapsed_time(e[2]) for e in r.stage_events])
