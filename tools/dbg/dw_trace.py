"""per-phase cycles of a k-step of the dW GEMM (library built with -DDINER_DW_TRACE): python tools/dbg/dw_trace.py path/to/lib.so"""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from diner_amd import _lib
_lib.LIB_PATH = Path(sys.argv[1]).resolve()
from diner_amd import training as T
dev = torch.device("cuda:0")
M = 655360
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((M, 512), device=dev, generator=g)
dY = torch.randn((M, 512), device=dev, generator=g) * 1e-3
dw = torch.zeros((512, 512), device=dev)
a = T.amax_of(dY, 1)
T.linear_bwd_w(dY, X, dw, None, relu_x=True, prec=1, amax=a)
torch.cuda.synchronize()
