#!/usr/bin/env python3
"""Timeline of one tile of the fused point/MLP kernel (diagnostic build, DINER_F16_TRACE=1): runs one cfg3-like launch, parses the
[f16 trace] lines and prints per wave the duration of every S phase (glue) and of every layer block (between its stamps).
    DINER_F16_TRACE=1 python tools/trace_f16.py 2> trace.txt ; python tools/trace_f16.py --parse trace.txt"""
import os, re, sys
import numpy as np

NAMES = {1: "tile", 10: "S->in", 11: "in|", 20: "S->net", 21: "net|", 30: "S->x", 31: "x|", 40: "S->head", 41: "head bar|", 42: "head red|"}

def parse(path):
    waves = {}
    for l in open(path):
        m = re.match(r"\[f16 trace\] wave(\d+):(.*)", l)
        if m:
            waves[int(m.group(1))] = [(int(a), int(b)) for a, b in (t.split(":") for t in m.group(2).split())]
    if not waves:
        print("no trace lines"); return
    t0 = min(ev[0][1] for ev in waves.values())
    for w, ev in sorted(waves.items()):
        s_time = blk_time = 0
        segs = []
        for (ida, ta), (idb, tb) in zip(ev[:-1], ev[1:]):
            d = tb - ta
            if idb in (10, 20, 30, 40):   # S phase ends at idb
                s_time += d; segs.append(f"S{d}")
            else:
                blk_time += d; segs.append(f"[{NAMES.get(idb, idb)}{d}]")
        tot = ev[-1][1] - ev[0][1]
        print(f"wave{w}: start {ev[0][1]-t0} total {tot} cycles  S {s_time} ({100*s_time/tot:.1f}%)  blocks {blk_time} ({100*blk_time/tot:.1f}%)")
        print("   " + " ".join(segs))

if "--parse" in sys.argv:
    parse(sys.argv[sys.argv.index("--parse") + 1]); sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"):
    from pathlib import Path
    _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
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
rays = torch.from_numpy(sc.target_rays()).to(dev)
import time
with torch.no_grad():
    r(m, rays); torch.cuda.synchronize()
    r.stage_events = []
    r(m, rays); torch.cuda.synchronize()
    e = r.stage_events[0]
    print("point/MLP kernel, full frame: %.1f ms" % e[1].elapsed_time(e[2]), file=sys.stderr)
