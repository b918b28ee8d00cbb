"""Hunt for a rare, timing-dependent mismatch (points_mlp_f16.hip, DINER_GEOM_WAVES): REPS fresh processes, each rendering goldens g1 / g3 / g0
through render_points twice per variant (lin_z as per-point GEMMs and as maps; the first renders of a process run with cold caches);
prints per render the number of samples beyond 1e-4 (@ray.sample of the first ones).  usage: [REPS=12] g1_race.py lib [lib ...]"""
import os, subprocess, sys
code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from pathlib import Path
from diner_amd import _lib
if os.environ.get("DINER_LIB_PATH"): _lib.LIB_PATH = Path(os.environ["DINER_LIB_PATH"]).resolve()
from tests.conftest import GoldenCase
from tests.test_gpu_parity import renderer_for, model_for, T
dev = torch.device("cuda:0")
res = []
for name, prec in (("g1_nv2_k64_dtu", "f16x3-gemm"), ("g3_nv3_k40_wide", "f16x3-gemm"), ("g0_nv4_k16", "f16x3-gemm"),
                   ("g1_nv2_k64_dtu", "f16x3"), ("g0_nv4_k16", "f16x3")):
    g = GoldenCase(name)
    r = renderer_for(g, prec); m = model_for(g, dev)
    with torch.no_grad():
        for i in range(2):
            o = r.render_points(m, T(g.rays, dev), T(g["z_fill"], dev)[None]).cpu().numpy()[0]
            d = np.abs(o[..., :3] - g["rgbsigma"][..., :3])
            bad = np.argwhere(d.max(-1) > 1e-4)
            nan = int((~np.isfinite(o)).any(-1).sum())   # (a poisoned tile: every sample NaN)
            res.append("%d%s%s" % (len(bad), ("@" + ",".join("%d.%d[%s]" % (b[0], b[1], "/".join("%.0e" % x for x in (o[b[0], b[1]] - g["rgbsigma"][b[0], b[1]]))) for b in bad[:3])) if len(bad) else "", ("!nan%d" % nan) if nan else ""))
print("RES", " ".join(res))
'''
for lib in sys.argv[1:]:
    for rep in range(int(os.environ.get("REPS", "4"))):
        env = dict(os.environ)
        if lib != "main": env["DINER_LIB_PATH"] = lib
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RES")]
        print(lib, rep, line[0] if line else p.stderr[-300:], flush=True)
