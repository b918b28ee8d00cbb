"""Compile-time guard for the register contract of the default point/MLP kernel (diner_amd/csrc/points_mlp_f16.hip).

Its GEMM core is generated assembly (diner_amd/csrc/gen_f16_core.py -> f16_core.inc) that OWNS the high VGPRs
v[CAP:255] (two accumulator grids + the weight ring, which stays in flight across barriers, layers and the compiler's glue
code).  That only works while hipcc keeps its own code below CAP, uses no AGPRs (at 2 waves per SIMD the budget is 256
registers in all) and never waits vmcnt(0) inside the core; checked on the generated ISA (cross-compilation, no GPU)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CSRC = ROOT / "diner_amd" / "csrc"


def _compile(tmp_path_factory, *defines):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    subprocess.run(["make", "-C", str(CSRC), "f16_core.inc", "f16_core16.inc", "f16_core16_trace.inc"], check=True, capture_output=True)
    asm = tmp_path_factory.mktemp("isa") / "points_mlp_f16.s"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-unroll-loops", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", *defines, "-o", str(asm), str(CSRC / "points_mlp_f16.hip")], check=True, capture_output=True, timeout=900)
    return asm.read_text()


@pytest.fixture(scope="module", params=["product", "diag"])
def isa(request, tmp_path_factory):
    """the product build (4 instantiations: lin_z maps or per-point lin_z GEMMs x view-sequential or views-in-tile) and the tools build
    with the TRACE instantiations (-DDINER_F16_DIAG: + 2)"""
    text = _compile(tmp_path_factory, *(["-DDINER_F16_DIAG"] if request.param == "diag" else []))
    return text, (6 if request.param == "diag" else 4)


def _cap():
    m = re.search(r"constexpr int F16_VGPR_CAP = (\d+);", (CSRC / "f16_core16.inc").read_text())
    return int(m.group(1))


def test_generated_core_is_current():
    """The generated cores in the tree are what the generator produces (the Makefile regenerates them; the files are committed so
    that the kernel sources read complete): f16_core16.inc (v_mfma_f32_16x16x32_f16: the inference kernel) with its stamped twin
    (TRACE instantiation only), and f16_core.inc (32x32x16: train_core.hip, tools/chain_probe.hip)."""
    for name, extra in (("f16_core16.inc", []), ("f16_core.inc", [])):
        text = (CSRC / name).read_text()
        args = re.search(r"// generator arguments: (.*)", text).group(1).split()
        assert ("--shape=16" in args) == (name == "f16_core16.inc")
        out = subprocess.run(["python3", str(CSRC / "gen_f16_core.py"), *args], check=True, capture_output=True, text=True).stdout
        assert out == text, name
    args = re.search(r"// generator arguments: (.*)", (CSRC / "f16_core16.inc").read_text()).group(1).split()
    out = subprocess.run(["python3", str(CSRC / "gen_f16_core.py"), *args, "--ns=tr", "--stamps"], check=True, capture_output=True, text=True).stdout
    assert out == (CSRC / "f16_core16_trace.inc").read_text()
    assert "v_mfma_f32_16x16x32_f16" in (CSRC / "f16_core16.inc").read_text() and "v_mfma_f32_32x32x16_f16" in (CSRC / "f16_core.inc").read_text()


def test_compiler_stays_out_of_the_core_registers(isa):
    isa, n_inst = isa
    cap = _cap()
    lines = isa.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5diner5f16x321points_mlp_f16_kernelILb[01]ELb[01]E\S+:", l)]
    assert len(starts) == n_inst, "expected <lin_z maps | per-point lin_z GEMMs> x <view-sequential | views-in-tile> (+ the two <trace> ones in the tools build)"
    for s0 in starts:
        end = next(i for i in range(s0, len(lines)) if "s_endpgm" in lines[i])
        in_asm, core_mfma, core_loads, stmt = False, 0, 0, []
        for i in range(s0, end):
            l = lines[i].split(";")[0] if not lines[i].lstrip().startswith(";;") else lines[i]
            if "#ASMSTART" in lines[i]:
                in_asm, stmt = True, []
                continue
            if "#ASMEND" in lines[i]:
                in_asm = False
                if any("v_mfma" in x for x in stmt):   # a layer block: its waits are counted, the weight ring never drains
                    assert not any("vmcnt(0)" in x for x in stmt), f"asm statement ending at line {i}: vmcnt(0) inside a GEMM block"
                continue
            if in_asm:
                stmt.append(l)
                core_mfma += "v_mfma" in l
                core_loads += "global_load_dwordx4" in l
                continue
            assert "v_accvgpr" not in l and not re.search(r"\ba\[?\d", l), f"line {i}: compiler-generated AGPR use: {l}"
            assert "v_mfma" not in l, f"line {i}: MFMA outside the generated core: {l}"
            for m in re.finditer(r"\bv\[?(\d+)(?::(\d+))?\]?", l):
                hi = int(m.group(2) or m.group(1))
                assert hi < cap, f"line {i}: compiler code touches v{hi} >= {cap} (the core's registers): {l}"
        assert core_mfma >= 400 and core_loads >= 100


def test_no_flat_instructions(isa):
    """No kernel of the library issues a FLAT instruction.  hipcc emits FLAT for every pointer it cannot type (one that passed through
    an empty asm; any volatile access).  A FLAT op goes down the LDS path and the memory path at once and counts in lgkmcnt as well as
    vmcnt.  With the view-sum slab on flat_store_dwordx4 a 4-wave geometry build returned rare wrong samples on the GPU -- 13-14 of 16
    processes of tools/dbg/g1_race.py with the stores flat, 0 of 16 with only the loads flat, 0 of 72 with global_store (DESIGN.md 4.1
    item 11: the mechanism is not established -- tools/flat_store_probe.hip does not reproduce it in isolation;
    tools/dbg/build_flat_repro.sh rebuilds the failing form at commit cbb8e8d).  Those accesses carry explicit address spaces now (points_mlp_f16.hip:
    gload4 / gstore4 / lds_vu32 / g_u32; train_core.hip)."""
    text, _ = isa
    flat = re.findall(r"^\s+(flat_\w+)", text, re.M)
    assert not flat, sorted(set(flat))


def test_no_flat_instructions_in_the_other_translation_units(tmp_path_factory):
    """... and the same for every other translation unit of libdiner_hip.so (compiled in parallel)"""
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_all")
    others = [p for p in sorted(CSRC.glob("*.hip")) if p.name != "points_mlp_f16.hip"]
    procs = [(p, subprocess.Popen([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-inline-asm", "-S",
                                   "--cuda-device-only", "-o", str(out / (p.stem + ".s")), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE))
             for p in others]
    for p, pr in procs:
        _, err = pr.communicate(timeout=1200)
        assert pr.returncode == 0, err.decode()[-400:]
        flat = re.findall(r"^\s+(flat_\w+)", (out / (p.stem + ".s")).read_text(), re.M)
        assert not flat, (p.name, sorted(set(flat)))


def test_register_budget(isa):
    """2 waves per SIMD: 256 registers per lane in all, none of them AGPRs."""
    isa, n_inst = isa
    meta = re.findall(r"\.agpr_count:\s+(\d+)\n\s+\.args:.*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", isa, flags=re.S)
    kern = [(int(a), n, int(v)) for a, n, v in meta if "points_mlp_f16_kernel" in n]
    assert len(kern) == n_inst
    for agpr, name, vgpr in kern:
        assert agpr == 0 and vgpr <= 256, (name, agpr, vgpr)


def test_product_library_carries_no_diagnostics():
    """the TRACE instantiation, its getenv / hipMalloc / fprintf live in the tools build only
    (tools/dbg/build_abl.sh trace -DDINER_F16_DIAG), not in libdiner_hip.so."""
    so = ROOT / "diner_amd" / "lib" / "libdiner_hip.so"
    if not so.exists():
        pytest.skip("library not built")
    blob = so.read_bytes()
    assert b"DINER_F16_TRACE" not in blob and b"[f16 trace]" not in blob
    assert b"points_mlp_f16_kernelILb1ELb1E" not in blob          # <LINZ, TRACE = true>
    assert b"points_mlp_f16_kernelILb1ELb0E" in blob and b"points_mlp_f16_kernelILb0ELb0E" in blob


def test_a_wait_that_gives_up_poisons_the_whole_tile():
    """the bounded spin of a flow-mode wait must end LOUD for the whole tile, not for one register of one wave: the
    generated timeout path sets the workgroup's poison word, and both kernels built on the core turn it into non-finite results."""
    for name in ("f16_core16.inc", "f16_core.inc", "f16_core16_trace.inc"):
        text = (CSRC / name).read_text()
        off = int(re.search(r"constexpr int F16_POISON_OFF = (\d+);", (CSRC / "f16_core16.inc").read_text()).group(1))
        waits = re.findall(r'"TW\d+_%=:\\n"\n(.*?)"D\d+_%=:\\n"', text, flags=re.S)
        assert len(waits) >= 9, name                                    # 3-4 waits per layer block, 3 blocks
        for w in waits:
            assert f"ds_write_b32 %[ctr], %[pv] offset:{off}" in w and "v_mov_b32 %[pv], 1" in w, name
    for src, what in (("points_mlp_f16.hip", "__builtin_nanf"), ("train_core.hip", "__builtin_nanf")):
        text = (CSRC / src).read_text()
        assert "LDS_CTR + F16_POISON_OFF" in text and what in text, src
    assert "amax = __builtin_inff()" in (CSRC / "train_core.hip").read_text()
