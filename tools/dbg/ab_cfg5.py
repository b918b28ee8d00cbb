"""A/B of libdiner_hip.so variants on the NV = 8 path: a quarter cfg5 frame (1024x1024 target, 8 views, K = 256; 262,144 rays) per library,
interleaved rounds, one process per measurement; prints the point/MLP kernel time.   usage: ab_cfg5.py main tools/dbg/libdiner_hip_X.so ..."""
import os, subprocess, sys
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from pathlib import Path
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"): _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
H = W = 1024; NV, K, G, NC = 8, 256, 96, 1000
sc = synth.make_scene(H, W, NV, seed=0, with_latent=False)
h, w = sc.latent_hw
latent = torch.randn((1, NV, 512, h, w), generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
m = model_from_scene(sc, synth.make_mlp_weights(7, bias_scale=0.1), device=dev, latent=latent)
r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G)
rays = torch.from_numpy(sc.target_rays()).to(dev)[:, :65536]
ts = []
with torch.no_grad():
    r(m, rays); torch.cuda.synchronize()
    for i in range(2):
        r.stage_events = []
        r(m, rays); torch.cuda.synchronize()
        e = r.stage_events[0]; ts.append(e[1].elapsed_time(e[2]))
print("RESULT min %.2f ms" % min(ts), ["%.1f" % t for t in ts])
'''
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "main": env["DINER_LIB_PATH"] = lib
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(rnd, lib, line[0] if line else p.stderr[-400:], flush=True)
