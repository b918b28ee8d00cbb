"""The reference's call granularity (64 calls of 4096 rays per 512 x 512 frame) under the three settings of the non-finite guard:
host-side cost of the guard per call ("rays_per_call_4096 not slower by > 0.5 %").  One process, interleaved rounds."""
import sys, time, torch
sys.path.insert(0, ".")
from diner_amd import NeRFRendererDGS
from synthetic import synth
from synthetic.model_stub import model_from_scene
dev = torch.device("cuda:0")
H = W = 512; NV, K, G, NC = 4, 128, 48, 1000
sc = synth.make_scene(H, W, NV, seed=0, with_latent=False)
h, w = sc.latent_hw
latent = torch.randn((1, NV, 512, h, w), generator=torch.Generator(device=dev).manual_seed(1234), device=dev)
m = model_from_scene(sc, synth.make_mlp_weights(7, bias_scale=0.1), device=dev, latent=latent)
rays = torch.from_numpy(sc.target_rays()).to(dev)
chunks = list(torch.split(rays, 4096, dim=1))
rs = {}
for mode in ("off", "deferred", "sync"):
    r = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G)
    r.finite_check = mode
    rs[mode] = r
with torch.no_grad():
    for r in rs.values():
        for ch in chunks[:4]:
            r(m, ch)
    torch.cuda.synchronize()
    for rnd in range(3):
        for mode, r in rs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ch in chunks:
                r(m, ch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"round {rnd} finite_check={mode:9s} {dt * 1e3:8.2f} ms per frame of 64 calls  {rays.shape[1] / dt:9.0f} rays/s", flush=True)
