#!/usr/bin/env python3
"""Ad-hoc: dX GEMM variants (diagnostic)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diner_amd import training as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M = 655360
X = torch.randn((M, 512), device=dev, generator=g)
W = torch.randn((512, 512), device=dev, generator=g) * 0.06
Wt = W.t().contiguous()
dY = torch.randn((M, 512), device=dev, generator=g) * 1e-3
out = torch.empty((M, 512), device=dev)
a = T.amax_of(dY, 1)
cases = {
    "dX mask amax": lambda: T.linear_bwd_x(dY, W, X, out, prec=1, amax=a),
    "dX nomask amax": lambda: T.linear_bwd_x(dY, W, None, out, prec=1, amax=a),
    "dX mask static": lambda: T.linear_bwd_x(dY, W, X, out, prec=1, amax=None),
    "dX as fwd(W^T) nomask": lambda: T.linear_fwd(dY, Wt, None, out, prec=1),
    "fwd": lambda: T.linear_fwd(X, W, None, out, relu_in=True, prec=1),
}
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"case {name:24s}: {ms:.3f} ms  {2.0*M*512*512/ms/1e9:.1f} TFLOP/s", flush=True)
