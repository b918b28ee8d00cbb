"""A/B of libdiner_hip.so variants on ONE box: for each library given ("main" = the in-tree build), time the point/MLP kernel on
half a cfg3 frame (131072 rays), interleaved rounds, one process per measurement.  A variant may carry environment switches:
`main@DINER_F16_NO_VIT=1` = the in-tree library with that variable set."""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pathlib import Path
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"): _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
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
r.finite_check = "off"      # (ablation variants produce garbage by design)
rays = torch.from_numpy(sc.target_rays()).to(dev)[:, :131072]
ts = []
with torch.no_grad():
    r(m, rays); torch.cuda.synchronize()
    for i in range(3):
        r.stage_events = []
        r(m, rays); torch.cuda.synchronize()
        e = r.stage_events[0]; ts.append(e[1].elapsed_time(e[2]))
print("RESULT min %.2f ms (frame-equivalent %.1f ms)" % (min(ts), 2 * min(ts)), ["%.1f" % t for t in ts])
'''
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        path, *sets = lib.split("@")
        for kv in sets:
            k, v = kv.split("=", 1)
            env[k] = v
        if path != "main":
            env["DINER_LIB_PATH"] = path
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(rnd, lib, line[0] if line else p.stderr[-400:], flush=True)
