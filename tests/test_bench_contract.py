"""The driver parses ONE JSON line from ``python bench.py``: check the contract (keys, types, the two extra objects) on the
small plumbing config so that an edit of bench.py cannot silently break it."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "tiny", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-rays", "256"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, t in dict(metric=str, value=float, unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=float,
                     higher_is_better=bool, scaling=str, dtype=str, data=str, config=dict, roofline=dict,
                     cpu_baseline=dict).items():
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["unit"] == "rays/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["rays_per_gpu_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert d["parity_on_sample"]["frac_rays_rgb_within_1e-4"] >= 0.97
