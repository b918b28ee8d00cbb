import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN_DIR = ROOT / "tests" / "golden"
GOLDEN_NAMES = sorted(p.stem for p in GOLDEN_DIR.glob("g[0-9]*.npz"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class GoldenCase:
    """A committed fixture + its inputs rebuilt from seeds (digest-checked)."""

    def __init__(self, name):
        from oracle.gen_golden import case_inputs, input_digests
        self.name = name
        self.data = dict(np.load(GOLDEN_DIR / f"{name}.npz", allow_pickle=False))
        self.cfg = json.loads(str(self.data["config"]))
        self.scene, self.weights, self.rays, self.noise = case_inputs(self.cfg)
        want = json.loads(str(self.data["digests"]))
        got = input_digests(self.scene, self.weights, self.rays, self.noise)
        assert got == want, f"seeded inputs of {name} drifted from the fixture: {got} != {want}"
        assert np.array_equal(self.rays, self.data["rays"])
        self.K, self.NC, self.G = self.cfg["K"], self.cfg["NC"], self.cfg["G"]

    def __getitem__(self, k):
        return self.data[k]


_cache = {}


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    if request.param not in _cache:
        _cache[request.param] = GoldenCase(request.param)
    return _cache[request.param]


def load_golden(name):
    if name not in _cache:
        _cache[name] = GoldenCase(name)
    return _cache[name]
