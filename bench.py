#!/usr/bin/env python3
"""Benchmark of the MI355X-native DINER render path: rendered rays/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--scaling auto|weak|strong]

A *step* renders one full target frame per GPU through the product path
(`diner_amd.NeRFRendererDGS.forward`: depth-guided sampler -> fused projection/gather/fusion-MLP
kernel -> alpha compositing), with maps, latent and weights already resident in HBM.

Multi-GPU: one process per GPU over RCCL.  `python bench.py --gpus N` as a plain
command starts its own N ranks (a fresh `python -m torch.distributed.run` child, spawned before this
process has touched the GPU) and relays rank 0's line; under an external torch.distributed.run
(RANK set) it is one of the ranks.  Two forms, both with the tile exchange inside the timed region:
  strong  ONE frame per step: its rays are split into N contiguous balanced ranges
          (`diner_amd.dist.shard_bounds`), every rank renders its range and the rendered [rays,4]
          tiles are all-gathered so that every rank holds the whole frame -- SURVEY.md §8(e)'s
          partitioning and the north_star question (">= 6x at 8 GPUs" on one 512x512 frame);
  weak    every rank renders its own target pose of the same scene: per-GPU work fixed, images
          sharded across GPUs first -- §8(e)'s form for cfg4 ("full val set").
`--scaling auto` (the default, what a bare `bench.py --gpus N` runs) is strong for cfg2 / cfg3 / cfg5 and
weak for cfg4; the OTHER form is then timed too and printed under `other_form` in the same line, and
`ranks_seen` lists what the process group really saw (world size, backend, every rank's device).

Rank 0 prints ONE JSON line (contract in the task description) with two extra objects:
  roofline      dominant kernel (fused point/MLP kernel, MFMA-bound): algorithmic FLOP per launch
                / its average duration from HIP events over the timed region, vs the dense MFMA
                peak of the arithmetic in use (fp16 for the default f16x3 mode, fp32 for
                --precision fp32); plus `sampling_integration` (HBM-bound kernels, logical bytes).
  cpu_baseline  the CPU oracle (a C port of the reference algorithm, oracle/) timed on the host
                cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (never the 2:1-sparsity marketing figures)
PEAK_MFMA_TFLOPS = {"fp32": 157.3, "f16x3": 2500.0}
MFMA_PASSES = {"fp32": 1, "f16x3": 3}   # f16x3: three fp16 MFMAs per fp32 product (hi*hi, hi*lo, lo*hi)
PEAK_HBM_GBS = 8000.0                   # HBM3E spec peak, same guide

CONFIGS = {
    # BASELINE.json configs[1]: DTU single scene 256x256, 4 views, 128 samples/ray (DTU near/far, black background)
    "cfg2": dict(H=256, W=256, NV=4, K=128, G=48, NC=1000, dataset="dtu", poses=8, cpu_rays=2048,
                 desc="cfg2: DTU-like scene, 256x256 target, 4 src views 256x256, K=128 (G=48), NC=1000"),
    # BASELINE.json configs[2] -- the configuration the metric is quoted on (512x512, 4 views, 128 samples)
    "cfg3": dict(H=512, W=512, NV=4, K=128, G=48, NC=1000, dataset="facescape", poses=8, cpu_rays=2048,
                 desc="cfg3: Facescape-like head, 512x512 target, 4 src views 512x512, K=128 (G=48), NC=1000, "
                      "depth-guided sampling on"),
    # BASELINE.json configs[3]: DTU val set 512x640 (non-square, configs/train_dtu.yaml:52-58, src/data/dtu.py:42-43),
    # 16 distinct target poses of the same scene stand in for the val images
    "cfg4": dict(H=512, W=640, NV=4, K=128, G=48, NC=1000, dataset="dtu", poses=16, cpu_rays=2048,
                 desc="cfg4: DTU-like scene, 512x640 target, 4 src views 512x640, K=128 (G=48), NC=1000, black background, "
                      "16 target poses"),
    # BASELINE.json configs[4] (stress): 1024x1024, 8 source views, 256 samples/ray; latent 576x576x512x8 = 5.4 GB,
    # lin_z maps 16 GB, 268 M points per frame
    "cfg5": dict(H=1024, W=1024, NV=8, K=256, G=96, NC=1000, dataset="facescape", poses=8, cpu_rays=384,
                 desc="cfg5: Facescape-like head, 1024x1024 target, 8 src views 1024x1024, K=256 (G=96), NC=1000"),
    # the cfg5 parameters at a quarter of its image size (a frame costs about as much as a cfg3 frame)
    "cfg5s": dict(H=256, W=256, NV=8, K=256, G=96, NC=1000, dataset="facescape", poses=8, cpu_rays=512,
                  desc="cfg5s: Facescape-like head, 256x256 target, 8 src views 256x256, K=256 (G=96), NC=1000"),
    # cfg4's shape (non-square, DTU, pose cycling) at plumbing size
    "cfg4s": dict(H=48, W=60, NV=4, K=64, G=24, NC=1000, dataset="dtu", poses=16, cpu_rays=256,
                  desc="cfg4s: cfg4's shape at 48x60 (plumbing only)"),
    # small plumbing config for quick checks
    "tiny": dict(H=64, W=64, NV=4, K=64, G=24, NC=1000, dataset="facescape", poses=8, cpu_rays=256,
                 desc="tiny: 64x64 target, 4 src views, K=64 (plumbing only)"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N>1: strong = one frame per step, its rays split over the GPUs (SURVEY 8(e); auto for cfg2/3/5); "
                         "weak = one frame per GPU per step (auto for cfg4: images first)")
    ap.add_argument("--no-other-form", action="store_true", help="N>1: do not also time the other scaling form")
    ap.add_argument("--rays-per-call", type=int, default=0, help="0 = whole frame in one launch (native mode); "
                    "4096 = the reference's ray_batch_size (src/models/diner.py:57)")
    ap.add_argument("--cpu-sample-rays", type=int, default=0, help="rays of the cpu_baseline sample (0 = the config's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline / exact-fp32 / 4096-ray legs")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32"], help="arithmetic of the fusion-MLP GEMMs")
    # CPU rehearsal of the launcher + sharding + gather + one-line contract (tests/test_bench_launch.py): gloo backend,
    # host tensors and a stub tile producer instead of the HIP render path.  Never a measurement.
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a fresh child
    (this process has not imported torch or touched HIP yet), relay rank 0's JSON line, return the child's status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + list(argv)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in p.stdout.splitlines():
        if l.startswith('{"metric"'):
            line = l
        elif l.strip():
            print(l, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return p.returncode


def default_form(config: str) -> str:
    """N>1 form of `--scaling auto` = SURVEY.md 8(e): one frame's rays split over the ranks ("strong") for cfg2 / cfg3 / cfg5,
    images first ("weak") for cfg4, whose unit of work is a validation set of images."""
    return "weak" if config.startswith("cfg4") else "strong"


def flops_per_ray(K, NV):
    """Algorithmic FLOPs of the fusion MLP per ray: K * 2 * (NV*2,387,456 + 1,050,624)."""
    return K * 2 * (NV * 2_387_456 + 1_050_624)


def host_cores():
    """CPU cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose all 256 hardware threads but grant 16 CPUs)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


KERNEL_SOURCES = ("points_mlp_f16.hip", "f16_core16.inc", "gen_f16_core.py", "points_mlp.hip", "common.hpp", "Makefile")


def _code_bytes(path: Path) -> bytes:
    """a source file as the digests see it: C++ sources without their `//` comments, trailing blanks and empty lines (no string literal
    of these files contains `//`), so that a comment-only edit does not orphan a recorded profile; everything else byte for byte"""
    data = path.read_bytes()
    if path.suffix not in (".hip", ".hpp", ".inc", ".h"):
        return data
    lines = (l.split("//", 1)[0].rstrip() for l in data.decode().splitlines())
    return "\n".join(l for l in lines if l).encode()


def kernel_source_digest() -> str:
    """sha256 (16 hex) over the sources the dominant kernel is built from (code only, see _code_bytes): ties a recorded PMC figure to a
    kernel version (.git does not travel to the GPU box, the sources do)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(_code_bytes(ROOT / "diner_amd" / "csrc" / name))
    return h.hexdigest()[:16]


SMALL_KERNEL_SOURCES = ("sampler.hip", "composite.hip", "common.hpp", "Makefile")


def small_kernel_source_digest() -> str:
    """the same for sampler_kernel / composite_kernel (the `sampling_integration` object)"""
    h = hashlib.sha256()
    for name in SMALL_KERNEL_SOURCES:
        h.update(_code_bytes(ROOT / "diner_amd" / "csrc" / name))
    return h.hexdigest()[:16]


def small_traffic_record(args):
    """measured L2-miss bytes per launch of the sampler and the compositing kernel + the sampler's VALU-busy fraction, recorded by
    tools/pmc_summary.py from the PMC passes of this command; None when never recorded or the two kernels changed since"""
    p = ROOT / "profiles" / "traffic.json"
    if not p.exists():
        return None
    rec = json.loads(p.read_text()).get(f"{args.config}:{args.precision}:{args.rays_per_call}:sampling_integration")
    if not isinstance(rec, dict) or rec.get("kernel_src_sha16") != small_kernel_source_digest():
        return None
    return rec


def traffic_record(args):
    """HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside the run (rocprofv3 collects them
    in separate passes), so this is the figure tools/pmc_summary.py recorded in profiles/traffic.json from such passes of this
    very command -- used only if the kernel sources still hash to what was profiled, else null.  Bytes = (2*FETCH_SIZE +
    WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md; they are L2-miss bytes, i.e. Infinity-Cache hits are included."""
    p = ROOT / "profiles" / "traffic.json"
    if not p.exists():
        return None, "no profiles/traffic.json"
    rec = json.loads(p.read_text()).get(f"{args.config}:{args.precision}:{args.rays_per_call}")
    if not isinstance(rec, dict):
        return None, "no PMC pass recorded for this config"
    if rec.get("kernel_src_sha16") != kernel_source_digest():
        return None, f"stale: {rec.get('source')} was taken with kernel sources {rec.get('kernel_src_sha16')}"
    return rec["bytes_per_launch"], rec.get("source")


def sampling_integration(args, b_s, b_c, rays_per_launch, t_samp, t_comp):
    """The north_star's ">= 40 % HBM roofline on the sampling + integration kernel" clause, stated honestly: `frac` is on SURVEY
    8(d)'s LOGICAL bytes (every map / tensor element counted once per access), as that clause defines it; next to it what the
    hardware moved (`traffic` = PMC L2-miss bytes of the two kernels, `frac_measured`) and what really bounds each kernel: the
    sampler is VALU-bound (its 33 MB of maps are cache-served), only the compositing kernel streams from HBM."""
    si_gbs = (b_s + b_c) * rays_per_launch / ((t_samp + t_comp) * 1e-3) / 1e9
    out = {"kernels": "sampler_kernel + composite_kernel", "bound": "hbm (logical bytes, SURVEY 8(d)); see `per_kernel` for the real bounds",
           "achieved": si_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": si_gbs / PEAK_HBM_GBS,
           "logical_bytes_per_ray": b_s + b_c, "avg_ms": [t_samp, t_comp], "traffic": None, "frac_measured": None,
           "per_kernel": {
               "sampler_kernel": {"bound": "valu", "logical_gbs": b_s * rays_per_launch / (t_samp * 1e-3) / 1e9, "avg_ms": t_samp},
               "composite_kernel": {"bound": "hbm", "logical_gbs": b_c * rays_per_launch / (t_comp * 1e-3) / 1e9, "avg_ms": t_comp,
                                    "frac": b_c * rays_per_launch / (t_comp * 1e-3) / 1e9 / PEAK_HBM_GBS}}}
    rec = small_traffic_record(args)
    if rec is None:
        out["traffic_source"] = "no PMC pass recorded for these kernel sources"
        return out
    tr = {k: rec[k]["bytes_per_launch"] for k in ("sampler", "composite") if k in rec}
    out["traffic"] = sum(tr.values())
    out["traffic_source"] = rec.get("source")
    out["frac_measured"] = out["traffic"] / ((t_samp + t_comp) * 1e-3) / 1e9 / PEAK_HBM_GBS
    pk = out["per_kernel"]
    if "sampler" in rec:
        pk["sampler_kernel"].update(traffic=tr["sampler"], measured_gbs=tr["sampler"] / (t_samp * 1e-3) / 1e9,
                                    valu_busy_frac=rec["sampler"].get("valu_busy_frac"))
    if "composite" in rec:
        pk["composite_kernel"].update(traffic=tr["composite"], measured_gbs=tr["composite"] / (t_comp * 1e-3) / 1e9,
                                      frac_measured=tr["composite"] / (t_comp * 1e-3) / 1e9 / PEAK_HBM_GBS)
    return out


def stub_tile(rays):
    """CPU rehearsal only (--stub-cpu): a deterministic function of the rays standing in for the render."""
    import torch
    r = rays[0]
    return torch.cat([r[:, 3:6] * 0.5 + 0.5, (r[:, 6:7] + r[:, 7:8]) * 0.5], 1).contiguous()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    # the hosts of this pool only support dmabuf IPC (RCCL / cross-process device memory): must be set before HIP starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    sys.path.insert(0, str(ROOT))
    from synthetic import synth
    from diner_amd.dist import OverlappedGather, all_gather_tiles, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    stub = args.stub_cpu
    if stub:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = CONFIGS[args.config]
    H, W, NV, K, G, NC = cfg["H"], cfg["W"], cfg["NV"], cfg["K"], cfg["G"], cfg["NC"]
    # N>1 default = SURVEY.md 8(e): one frame's rays split over the ranks (cfg2 / cfg3 / cfg5), images first for cfg4
    images_first = args.config.startswith("cfg4")     # (also: its steps cycle through the config's target poses)
    strong = world > 1 and (args.scaling if args.scaling != "auto" else default_form(args.config)) == "strong"

    # what the process group really is: world size and backend from torch.distributed, every rank's device gathered to all
    me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(),
          "device_index": None if stub else torch.cuda.current_device(),
          "device_name": "cpu (stub)" if stub else torch.cuda.get_device_name(dev)}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": seen}
    else:
        ranks_seen = {"world_size": 1, "backend": None, "ranks": [me]}

    # ---- synthetic scene, resident in HBM before the timed region ------------
    scene = synth.make_scene(H, W, NV, seed=0, dataset=cfg["dataset"], with_latent=False)
    h, w = scene.latent_hw
    rend = model = latent = weights = None
    if not stub:
        from diner_amd import NeRFRendererDGS
        from synthetic.model_stub import model_from_scene
        gen = torch.Generator(device=dev).manual_seed(1234)
        latent = torch.randn((1, NV, 512, h, w), generator=gen, device=dev, dtype=torch.float32)
        weights = synth.make_mlp_weights(7, bias_scale=0.1)  # seed with sigma > 0 almost everywhere: a meaningful parity sample
        model = model_from_scene(scene, weights, device=dev, latent=latent)
        rend = NeRFRendererDGS(n_samples=K, n_depth_candidates=NC, n_gaussian=G, white_bkgd=scene.white_bkgd)
        rend.precision = args.precision
    # target poses: weak = every rank renders its own pose in every step; strong = all ranks share the step's pose
    n_poses = cfg["poses"]
    yaws = np.linspace(-0.4, 0.45, n_poses) if images_first else 0.1 + 0.07 * np.arange(n_poses)

    def rays_of_pose(pi):
        scene.target_extrinsics = synth.look_at_origin_w2c(float(yaws[pi % n_poses]), scene.meta["cam_radius"])
        return torch.from_numpy(scene.target_rays()).to(dev)  # [1, H*W, 8]

    NR = H * W

    def fence():
        if not stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    class Form:
        """One scaling form of the step: which rays this rank renders, the double-buffered all-gather of the [rays,4] tiles."""

        def __init__(self, strong):
            self.strong = strong
            self.lo, self.hi = shard_bounds(NR, world, rank) if strong else (0, NR)
            self.n_mine = self.hi - self.lo
            self.poses = sorted({self.pose_of_step(i) for i in range(args.steps + args.warmup)})
            self.rays = {pi: rays_of_pose(pi)[:, self.lo:self.hi].contiguous() for pi in self.poses}
            self.rpc = args.rays_per_call if args.rays_per_call > 0 else self.n_mine
            self.n_gathered = NR if strong else world * NR
            # ONE all-gather per frame, double-buffered: it travels while the next frame renders
            self.og = OverlappedGather(self.n_gathered, world, rank, 4, dev)
            assert self.og.n_mine == self.n_mine

        def pose_of_step(self, i, r=rank):
            if not images_first:  # one fixed pose per rank (weak) / one pose (strong), as in round 1
                return 0 if self.strong else r % n_poses
            return (i if self.strong else i * world + r) % n_poses

        def render_into_tile(self, rays):
            tile = self.og.tile()
            o = 0
            for ch in torch.split(rays, self.rpc, dim=1):
                n = ch.shape[1]
                if stub:
                    tile[o:o + n] = stub_tile(ch)
                else:
                    out = rend(model, ch)
                    tile[o:o + n, :3] = out.fine.rgb[0]
                    tile[o:o + n, 3] = out.fine.depth[0]
                o += n

        def step(self, i):
            self.render_into_tile(self.rays[self.pose_of_step(i)])
            return self.og.submit()  # starts this frame's all-gather (RCCL), hands back the previous frame

        def timed(self):
            """W warm-up steps, then exactly K steps between two fences; max over ranks; -> (seconds, last frame, stage events)"""
            with torch.no_grad():
                for i in range(args.warmup):
                    self.step(i)
                self.og.flush()
                if rend is not None:
                    rend.stage_events = []
                fence()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    self.step(args.warmup + i)
                frame = self.og.flush()          # the last frame's gather is inside the timed region
                fence()
                elapsed = time.perf_counter() - t0
                events = []
                if rend is not None:
                    events, rend.stage_events = rend.stage_events, None
            if world > 1:
                t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            assert frame.shape[0] == self.n_gathered
            return elapsed, frame, events

        def stub_frame_ok(self, frame):
            """rehearsal: the gathered frame must be what one process would have produced"""
            last = args.warmup + args.steps - 1
            if self.strong:
                want = stub_tile(rays_of_pose(self.pose_of_step(last)))
            else:
                want = torch.cat([stub_tile(rays_of_pose(self.pose_of_step(last, r))) for r in range(world)], 0)
            return bool(torch.equal(frame, want))

        def rays_per_step(self):
            return NR if self.strong else world * NR     # rays all ranks rendered per step

    form = Form(strong)
    elapsed, frame, events = form.timed()
    stub_ok = form.stub_frame_ok(frame) if stub else None
    n_mine, rpc = form.n_mine, form.rpc
    render_into_tile, og = form.render_into_tile, form.og
    rays_by_pose, my_poses = form.rays, form.poses
    other = None
    if world > 1 and not args.no_other_form:       # the other form, same steps, printed beside the headline one
        of = Form(not strong)
        o_elapsed, o_frame, _ = of.timed()
        other = {"scaling": "strong" if of.strong else "weak", "value": of.rays_per_step() * args.steps / o_elapsed, "unit": "rays/s",
                 "ms_per_step": o_elapsed / args.steps * 1e3, "rays_per_gpu_per_step": of.n_mine}
        if stub:
            other["stub_frame_ok"] = of.stub_frame_ok(o_frame)
        del of, o_frame

    rays_per_step = form.rays_per_step()
    result = {
        "metric": "rendered rays/sec (512x512, 4 src views, 128 samples/ray)",
        "value": rays_per_step * args.steps / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "other_form": other,
        "ranks_seen": ranks_seen,
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 (GEMM operands split into fp16 hi+lo, 3 fp16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic" if not stub else "stub (CPU rehearsal of the launcher, not a measurement)",
        "config": {"workload": cfg["desc"], "rays_per_gpu_per_step": n_mine, "rays_per_call": rpc,
                   "source_views": NV, "samples_per_ray": K, "n_gaussian": G, "n_candidates": NC,
                   "parallelism": ("single GPU" if world == 1 else
                                   f"one frame's rays split into {world} contiguous ranges, RCCL all-gather of [rays,4] tiles" if strong else
                                   f"rays sharded x{world} (one target frame per GPU), RCCL all-gather of [rays,4] tiles")},
    }

    if not stub:
        # ---- per-kernel durations from the HIP events recorded inside the timed region --------------
        ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in events])  # [launches, 3]
        t_samp, t_mlp, t_comp = [float(x) for x in ms.mean(0)]
        rays_per_launch = min(rpc, n_mine)
        f_launch = flops_per_ray(K, NV) * rays_per_launch          # algorithmic FLOP of the reference's MLP
        achieved_tflops = f_launch / (t_mlp * 1e-3) / 1e12
        # what the matrix cores actually execute: with the lin_z maps, 3 of the 9 per-view 512x512 layers are
        # pre-multiplied once per encode (diner_pack_linz_maps) and leave the per-point kernel; the last per-view fc_1
        # is applied once to the view mean (it commutes with the mean)
        linz = args.precision == "f16x3" and rend.linz_maps
        macs = NV * (2_387_456 - (3 * 512 * 512 if linz else 0)) + 1_050_624
        if args.precision == "f16x3" and getattr(rend, "fc1_on_mean", False):
            macs -= (NV - 1) * 512 * 512
        f_exec = K * 2 * macs * rays_per_launch * MFMA_PASSES[args.precision]
        peak = PEAK_MFMA_TFLOPS[args.precision]
        b_s, b_c = 32 + NC * NV * 20 + 4 * K, K * 20 + 32 + 16  # logical bytes/ray
        traffic, traffic_source = traffic_record(args)
        result["roofline"] = {
            "kernel": "points_mlp_kernel" if args.precision == "fp32" else "points_mlp_f16_kernel",
            "bound": "mfma", "achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved_tflops / peak, "traffic": traffic, "traffic_source": traffic_source,
            "flop_per_launch": f_launch, "avg_ms": t_mlp,
            "mfma_dtype": "f32" if args.precision == "fp32" else "f16",
            "executed_tflops": f_exec / (t_mlp * 1e-3) / 1e12,
            "executed_frac": f_exec / (t_mlp * 1e-3) / 1e12 / peak,
            "note": "achieved = algorithmic FLOP of the reference MLP / kernel time; executed = MFMA FLOP actually "
                    "issued (x3 passes in f16x3 mode, minus the layers hoisted out of the per-point path); traffic = L2-miss "
                    "bytes (HBM + Infinity Cache) of a recorded PMC pass of this command, null when the kernel changed since",
            "sampling_integration": sampling_integration(args, b_s, b_c, rays_per_launch, t_samp, t_comp)}
    else:
        result["roofline"] = None
        result["stub_frame_ok"] = stub_ok

    extras = rank == 0 and world == 1 and not args.no_cpu_baseline and not stub
    if extras:
        rays = rays_by_pose[my_poses[0]]

    # ---- the same frame with exact fp32 MFMA (v_mfma_f32_32x32x2_f32) for reference, N=1 cfg3/cfg2/tiny only: the
    #      default f16x3 mode is an fp32-grade emulation (parity tests hold both modes to the same bars), this shows
    #      what the emulation buys and that nothing hides behind it
    if extras and args.precision == "f16x3" and NR * K * NV <= 512 * 512 * 128 * 4:
        rend.precision = "fp32"
        with torch.no_grad():
            render_into_tile(rays)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            render_into_tile(rays)
            torch.cuda.synchronize()
            t_fp32 = time.perf_counter() - t0
        rend.precision = args.precision
        result["exact_fp32_mfma"] = {"value": NR / t_fp32, "unit": "rays/s", "ms_per_step": t_fp32 * 1e3,
                                     "frac_of_fp32_mfma_peak": f_launch / t_fp32 / 1e12 / PEAK_MFMA_TFLOPS["fp32"]}

    # ---- the same frame in the reference's call granularity (ray_batch_size 4096, src/models/diner.py:57): SURVEY.md
    #      §8(d) asks for both the whole-frame launch (`value`) and this one
    if extras and rpc == NR and 4096 < NR <= 512 * 640:
        small = list(torch.split(rays, 4096, dim=1))
        with torch.no_grad():
            for ch in small[:4]:
                rend(model, ch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o, tile = 0, og.tile()
            for ch in small:
                out = rend(model, ch)
                n = ch.shape[1]
                tile[o:o + n, :3] = out.fine.rgb[0]
                tile[o:o + n, 3] = out.fine.depth[0]
                o += n
            torch.cuda.synchronize()
            t_small = time.perf_counter() - t0
        result["rays_per_call_4096"] = {"value": NR / t_small, "unit": "rays/s", "ms_per_frame": t_small * 1e3, "calls": len(small)}

    # ---- CPU baseline: the oracle (C port of the reference algorithm) on a bounded sample ---------
    if extras:
        from oracle.oracle import Oracle
        cores = host_cores()
        n_s = min(args.cpu_sample_rays or cfg["cpu_rays"], NR)
        sel = np.linspace(0, NR - 1, n_s).astype(np.int64)
        rays_s = np.ascontiguousarray(rays.cpu().numpy()[:, sel])
        scene.latent = latent.cpu().numpy()
        orc = Oracle(scene, weights, threads=cores)
        noise = synth.make_noise(n_s, NC, G, K, seed=5)
        t0 = time.perf_counter()
        ref = orc.render(rays_s, NC, K, G, noise, white_bkgd=scene.white_bkgd)
        t_cpu = time.perf_counter() - t0
        with torch.no_grad():
            out = rend(model, torch.from_numpy(rays_s).to(dev), noise=tuple(torch.from_numpy(n).to(dev)[None] for n in noise))
        d = np.abs(out.fine.rgb.cpu().numpy()[0] - ref["rgb"]).max(-1)
        result["cpu_baseline"] = {"value": n_s / t_cpu, "unit": "rays/s", "cores": cores, "kind": "port",
                                  "sample": f"{n_s} rays strided over the same frame/config, full path "
                                            f"(sampler+MLP+compositing), OpenMP over {cores} threads, {t_cpu:.1f} s"}
        result["parity_on_sample"] = {"rays": int(n_s), "frac_rays_rgb_within_1e-4": float((d <= 1e-4).mean()),
                                      "max_abs_rgb_diff": float(d.max()), "median_abs_rgb_diff": float(np.median(d))}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
