import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from diner_amd import _lib
if os.environ.get("DINER_LIB") == "v1" or os.environ.get("DINER_LIB_PATH"):
    from pathlib import Path
    _lib.LIB_PATH = Path(os.environ.get("DINER_LIB_PATH", "tools/dbg/libdiner_hip_v1.so")).resolve()
from tests.conftest import load_golden
from tests.test_gpu_parity import renderer_for, model_for, T
dev = torch.device("cuda:0")
print("library:", _lib.LIB_PATH)
for name in (sys.argv[1:] or ["g3_nv3_k40_wide"]):
    g = load_golden(name)
    ref = g["rgbsigma"]
    outs = {}
    for prec in ("fp32", "f16x3", "f16x3-gemm"):
        r = renderer_for(g, prec); m = model_for(g, dev)
        runs = []
        for rep in range(3):
            with torch.no_grad():
                runs.append(r.render_points(m, T(g.rays, dev), T(g["z_fill"], dev)[None]).cpu().numpy()[0])
        outs[prec] = runs[0]
        nd = [int((runs[0] != x).sum()) for x in runs[1:]]
        d = np.abs(outs[prec] - ref)
        bad = np.argwhere(d[..., :3] > 5e-5)
        print(name, prec, "max rgb diff %.3e" % d[..., :3].max(), "max sigma diff %.3e" % d[..., 3].max(), "n>5e-5:", len(bad), "run-to-run differing values:", nd)
        for b in bad[:10]:
            print("   ray %d sample %d ch %d: got %.6f ref %.6f fp32 %.6f | sigma got %.4f ref %.4f" % (b[0], b[1], b[2], outs[prec][tuple(b)], ref[tuple(b)], outs["fp32"][tuple(b)], outs[prec][b[0], b[1], 3], ref[b[0], b[1], 3]))
