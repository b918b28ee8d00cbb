"""only the dW GEMM of the training path (f16x3), for PMC passes: python tools/dbg/dw_only.py"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from diner_amd import training as T
dev = torch.device("cuda:0")
M = 655360
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((M, 512), device=dev, generator=g)
dY = torch.randn((M, 512), device=dev, generator=g) * 1e-3
dw = torch.zeros((512, 512), device=dev)
a = T.amax_of(dY, 1)
for _ in range(3):
    T.linear_bwd_w(dY, X, dw, None, relu_x=True, prec=1, amax=a)
torch.cuda.synchronize()
print("done")
