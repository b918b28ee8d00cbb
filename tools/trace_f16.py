#!/usr/bin/env python3
"""Timeline of one tile of the fused point/MLP kernel (diagnostic build, DINER_F16_TRACE=1): runs one cfg3-like launch, parses the
[f16 trace] lines and prints per wave and layer block the glue time before it and the five phases inside it (waits on the arrival counters, the two GEMM halves).
    DINER_F16_TRACE=1 python tools/trace_f16.py 2> trace.txt ; python tools/trace_f16.py --parse trace.txt"""
import os, re, sys
import numpy as np

def parse(path):
    """per layer block of a wave: S (glue before the block) | waitA (region 0 ready) | G1 | waitB (region 1 ready) | G2 | waitF (operand
    rows free); events 100..105 are the stamps inside the generated block (f16_core_trace.inc), 10/20/30 name the block that follows."""
    waves = {}
    for l in open(path):
        m = re.match(r"\[f16 trace\] wave(\d+):(.*)", l)
        if m:
            # stamps are relative to wave 0's first one, printed as unsigned 64-bit: a wave that started earlier shows a wrapped value
            waves[int(m.group(1))] = [(int(a), int(b) - (1 << 64) if int(b) >= (1 << 63) else int(b)) for a, b in (t.split(":") for t in m.group(2).split())]
    if not waves:
        print("no trace lines"); return
    for w, ev in sorted(waves.items()):
        print(f"wave{w}: (start {ev[0][1]})")
        prev_end, i, last_block = ev[0][1], 0, "?"
        tot = dict(S=0, waitA=0, G1=0, waitB=0, G2=0, waitF=0)
        while i < len(ev):
            if ev[i][0] in (10, 20, 30):
                last_block = {10: "in ", 20: "net", 30: "x  "}[ev[i][0]]
            if ev[i][0] == 100 and i + 5 < len(ev):
                t = [ev[i + k][1] for k in range(6)]
                d = [t[0] - prev_end] + [t[k + 1] - t[k] for k in range(5)]
                print(f"  {last_block} @{t[0]:>8}  " + "  ".join(f"{k} {v:>6}" for k, v in zip(tot, d)))
                for k, v in zip(tot, d):
                    tot[k] += v
                prev_end = t[5]
                i += 6
            else:
                i += 1
        T = max(1, sum(tot.values()))
        print("  totals: " + "  ".join(f"{k} {v} ({100 * v / T:.1f}%)" for k, v in tot.items()))

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
