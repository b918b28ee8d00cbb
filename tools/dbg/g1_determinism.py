"""Debug: golden g1 (NV = 2, K = 64, DTU) through render_points, lin_z as per-point GEMMs (the 'f16x3-gemm' variant), repeated: is the
output deterministic run to run, which samples differ from the golden, and by how much.  DINER_LIB_PATH selects a library variant."""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from pathlib import Path
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"): _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
from tests.conftest import GoldenCase
from tests.test_gpu_parity import renderer_for, model_for, T
dev = torch.device("cuda:0")
g = GoldenCase("g1_nv2_k64_dtu")
for prec in ("f16x3-gemm", "f16x3"):
    r = renderer_for(g, prec)
    m = model_for(g, dev)
    outs = []
    with torch.no_grad():
        for i in range(6):
            outs.append(r.render_points(m, T(g.rays, dev), T(g["z_fill"], dev)[None]).cpu().numpy()[0])
    ref = g["rgbsigma"]
    for i, o in enumerate(outs):
        d = np.abs(o[..., :3] - ref[..., :3])
        bad = np.argwhere(d > 1e-4)
        same = np.array_equal(o, outs[0])
        print(prec, "run", i, "identical to run 0:", same, "| samples beyond 1e-4:", len(bad), "max", float(d.max()),
              "at", np.unravel_index(d.argmax(), d.shape), flush=True)
    if not all(np.array_equal(o, outs[0]) for o in outs):
        dd = np.abs(outs[1] - outs[0]); print("  run-to-run max diff", float(dd.max()), "count", int((dd > 0).sum()))
