import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from diner_amd import _lib
if os.environ.get("DINER_LIB") == "v1":
    from pathlib import Path
    _lib.LIB_PATH = Path("tools/dbg/libdiner_hip_v1.so").resolve()
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
print("library:", _lib.LIB_PATH.name)
for NV in (2, 3, 4, 5, 6, 8):
    for fs in (1.0, 0.45):
        sc = synth.make_scene(32, 32, NV, seed=9, feature_padding=4)
        w = synth.make_mlp_weights(10, bias_scale=0.1)
        rays = sc.target_rays(focal_scale=fs)[:, ::8]
        K = 40
        z = np.sort(np.random.RandomState(1).uniform(sc.near, sc.far, (1, rays.shape[1], K)).astype(np.float32), -1)
        m = model_from_scene(sc, w, device=dev)
        out = {}
        for prec in ("fp32", "f16x3"):
            r = NeRFRendererDGS(n_samples=K, n_depth_candidates=64, n_gaussian=4)
            r.precision = prec
            with torch.no_grad():
                out[prec] = r.render_points(m, T(rays), T(z)).cpu().numpy()[0]
        d = np.abs(out["f16x3"] - out["fp32"])
        print(f"NV={NV} focal_scale={fs}: f16x3 vs fp32 kernel max rgb {d[..., :3].max():.2e} p99.9 {np.percentile(d[..., :3], 99.9):.2e} max sigma {d[..., 3].max():.2e}")
