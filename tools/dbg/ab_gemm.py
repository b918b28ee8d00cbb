"""A/B of libdiner_hip.so variants for the training core GEMM: python tools/dbg/ab_gemm.py main path/to/variant.so"""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, ".")
from pathlib import Path
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"): _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
from diner_amd import training as T
dev = torch.device("cuda:0"); M = 655360
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((M, 512), device=dev, generator=g); W = torch.randn((512, 512), device=dev, generator=g) * 0.06
b = torch.zeros(512, device=dev); out = torch.empty((M, 512), device=dev)
wc = T.split_panel(W, False, 1)
res = []
for name, kw in (("fwd", {}), ("fwd+add", dict(addend=out))):
    fn = lambda: T.linear_fwd(X, W, b, out, relu_in=True, prec=1, panel=wc, **kw)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    res.append("%s %.3f ms" % (name, e0.elapsed_time(e1) / 5))
print("RESULT", "  ".join(res))
'''
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "main": env["DINER_LIB_PATH"] = lib
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(rnd, lib, line[0] if line else p.stderr[-300:], flush=True)
