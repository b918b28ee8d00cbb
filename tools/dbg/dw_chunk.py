"""dW GEMM time against the split-K chunk: python tools/dbg/dw_chunk.py"""
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
for kc in (1024, 2048, 4096, 8192, 16384, 32768):
    T.K_CHUNK = kc
    fn = lambda: T.linear_bwd_w(dY, X, dw, None, relu_x=True, prec=1, amax=a)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"K_CHUNK {kc:6d}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
