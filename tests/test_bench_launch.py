"""`python bench.py --gpus N` as a plain command (how the driver runs it): bench.py must start its own ranks, shard,
gather and print ONE contract line.  Rehearsed on CPU: world 2 over gloo with the stub tile producer (`--stub-cpu`), both
scaling forms; the sharding / gather / launcher code is exactly what the GPU run uses."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def run(*extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), "--stub-cpu", "--config", "tiny", "--steps", "2", "--warmup", "1",
                           *extra], capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=e)


@pytest.mark.parametrize("scaling,config", [("weak", "tiny"), ("strong", "tiny"), ("weak", "cfg4s"), ("strong", "cfg4s")])
def test_self_launch_world2(scaling, config):
    extra = ["--gpus", "2", "--scaling", scaling]
    if config != "tiny":
        extra += ["--config", config]
    p = run(*extra)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2 and d["warmup"] == 1
    assert d["stub_frame_ok"] is True
    n = d["config"]["rays_per_gpu_per_step"]
    H, W = (64, 64) if config == "tiny" else (48, 60)
    assert n == (H * W // 2 if scaling == "strong" else H * W)
    total = H * W if scaling == "strong" else 2 * H * W
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    # the other form is timed too and printed beside the headline one
    o = d["other_form"]
    assert o["scaling"] == ("weak" if scaling == "strong" else "strong") and o["stub_frame_ok"] is True
    assert o["rays_per_gpu_per_step"] == (H * W if scaling == "strong" else H * W // 2)
    o_total = 2 * H * W if scaling == "strong" else H * W
    assert abs(o["value"] - o_total / (o["ms_per_step"] * 1e-3)) <= 1e-6 * o["value"]
    # what the process group really saw
    rs = d["ranks_seen"]
    assert rs["world_size"] == 2 and rs["backend"] == "gloo" and [r["rank"] for r in rs["ranks"]] == [0, 1]
    assert all("device_name" in r and "local_rank" in r for r in rs["ranks"])


@pytest.mark.parametrize("config,want", [("tiny", "strong"), ("cfg4s", "weak")])
def test_default_form_is_survey_8e(config, want):
    """A bare `bench.py --gpus N` (how the driver runs it) answers north_star's question: one frame's rays split into N contiguous
    ranges + one all-gather for the square-frame configs (cfg2 / cfg3 / cfg5 and their plumbing stand-in `tiny`), images first
    for cfg4 (here its plumbing-size twin cfg4s) -- SURVEY.md 8(e); the loop being sharded is the reference's src/models/diner.py:85-92."""
    extra = ["--gpus", "2"] + (["--config", config] if config != "tiny" else [])
    p = run(*extra)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    H, W = (64, 64) if config == "tiny" else (48, 60)
    assert d["scaling"] == want and d["stub_frame_ok"] is True
    assert d["config"]["rays_per_gpu_per_step"] == (H * W // 2 if want == "strong" else H * W)
    assert d["other_form"]["scaling"] != want and d["ranks_seen"]["world_size"] == 2


def test_default_form_table():
    """the auto rule for every BASELINE configuration, without launching anything"""
    sys.path.insert(0, str(ROOT))
    import bench
    for cfg, want in dict(cfg2="strong", cfg3="strong", cfg5="strong", cfg4="weak").items():
        a = bench.parse_args(["--gpus", "8", "--config", cfg])
        assert a.scaling == "auto" and bench.default_form(a.config) == want


def test_world_mismatch_is_an_error():
    p = run("--gpus", "2", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_single_process_stub():
    p = run()
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip())
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["stub_frame_ok"] is True
    assert d["other_form"] is None and d["ranks_seen"]["world_size"] == 1


def test_traffic_figure_is_tied_to_the_kernel_sources(monkeypatch):
    """roofline.traffic is a recorded PMC figure: quoted only while the kernel sources hash to what was profiled."""
    import argparse
    import importlib
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    a = argparse.Namespace(config="cfg3", precision="f16x3", rays_per_call=0)
    val, src = bench.traffic_record(a)
    rec = json.loads((ROOT / "profiles" / "traffic.json").read_text())["cfg3:f16x3:0"]
    if rec["kernel_src_sha16"] == bench.kernel_source_digest():
        assert val == rec["bytes_per_launch"] and "pmc_cfg3" in src
    else:
        assert val is None and src.startswith("stale")
    monkeypatch.setattr(bench, "kernel_source_digest", lambda: "0" * 16)
    val, src = bench.traffic_record(a)
    assert val is None and src.startswith("stale")
    val, src = bench.traffic_record(argparse.Namespace(config="tiny", precision="f16x3", rays_per_call=0))
    assert val is None


def test_source_digest_sees_code_not_comments(tmp_path):
    """the digest that ties a profile to a kernel version ignores `//` comments and blank lines of the C++ sources (a comment-only edit
    must not orphan a recorded profile) and nothing else"""
    import importlib
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    a, b, c, py = tmp_path / "a.hip", tmp_path / "b.hip", tmp_path / "c.hip", tmp_path / "g.py"
    a.write_text("// header\nint f() { return 1; }   // one\n\n")
    b.write_text("int f() { return 1; }\n// another comment\n")
    c.write_text("int f() { return 2; }\n")
    py.write_text("x = 1  # // not C++\n")
    assert bench._code_bytes(a) == bench._code_bytes(b) != bench._code_bytes(c)
    assert bench._code_bytes(py) == py.read_bytes()
