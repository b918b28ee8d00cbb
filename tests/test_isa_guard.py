"""Compile-time guard for the weight-stream ring of the default point/MLP kernel (diner_amd/csrc/points_mlp_f16.hip).

The ring is written with inline-asm loads and counted ``s_waitcnt vmcnt(4)``; it only prefetches if the compiler
keeps spill code out of the k-loops: a ``scratch_load`` there is followed by a compiler-inserted ``vmcnt(0)``
that drains the ring every step (seen in the diagnostic STAMP build, ~10 % slower).  hipcc's register allocation
of this 256-VGPR kernel is sensitive to unrelated edits, so the property is checked on the generated ISA
(cross-compilation, no GPU needed)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_f16_gemm_loops_have_no_spill_code(tmp_path):
    asm = tmp_path / "points_mlp_f16.s"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-unroll-loops", "-S", "--cuda-device-only",
                    "-o", str(asm), str(ROOT / "diner_amd/csrc/points_mlp_f16.hip")], check=True, capture_output=True, timeout=900)
    lines = asm.read_text().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5diner5f16x321points_mlp_f16_kernelILb0E\S+:", l)]
    assert len(starts) == 2, "expected the two production instantiations <false,true> and <false,false>"
    for s0 in starts:
        end = next(i for i in range(s0, len(lines)) if "s_endpgm" in lines[i])
        labels, loops = {}, []
        for i in range(s0, end):
            m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
            if m:
                labels[m.group(1)] = i
            m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", lines[i])
            if m and m.group(1) in labels:
                loops.append((labels[m.group(1)], i))
        inner = [(a, b) for a, b in loops if sum("v_mfma" in l for l in lines[a:b]) == 24]  # one ring turn = 2 k-blocks x 12 MFMAs
        assert len(inner) >= 4, f"k-loops not found ({len(inner)})"
        for a, b in inner:
            body = lines[a:b]
            assert sum("vmcnt(4)" in l for l in body) == 2, "ring waits missing"
            assert not any("scratch_" in l for l in body), f"spill code inside a GEMM k-loop (lines {a}-{b})"
            assert not any("vmcnt(0)" in l for l in body), f"vmcnt(0) inside a GEMM k-loop (lines {a}-{b})"
