#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/prof.sh into profiles/:

    python tools/pmc_summary.py cfg3 [--round r02] [--precision f16x3] [--rays-per-call 0]

  profiles/<round>/kernel_stats_<cfg>_<precision>.csv   per-kernel calls / total / average duration (kernel-trace pass)
  profiles/<round>/pmc_<cfg>_<precision>.json            counters per kernel, averaged over the launches of the pass
  profiles/traffic.json                                  HBM(+Infinity Cache) bytes per launch of the dominant kernel =
                                                         (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reads half the bytes of wide loads,
                                                         MI355X_MICROARCH.md), tagged with the sha256 of the kernel sources it was measured on:
                                                         bench.py only quotes it while the sources still hash to that value
"""
import argparse, csv, glob, json, os, sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--round", default="r03")
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--rays-per-call", type=int, default=0)
    a = ap.parse_args()
    src = ROOT / "gpurun_out" / f"prof_{a.config}"
    out = ROOT / "profiles" / a.round
    out.mkdir(parents=True, exist_ok=True)
    tag = f"{a.config}_{a.precision}"
    # ---- kernel stats
    ks = sorted(glob.glob(str(src / "kt" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run of the pass
    if ks:
        rows = list(csv.DictReader(open(ks[-1])))
        with open(out / f"kernel_stats_{tag}.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print("wrote", out / f"kernel_stats_{tag}.csv")
    # ---- counters
    # a dispatch reports one row per counter instance: sum the instances of a dispatch, average over dispatches
    res = {}
    newest = {}   # gpurun merges every run's files into the same local directory: per pass keep the newest run only
    for p in glob.glob(str(src / "*" / "*" / "*_counter_collection.csv")):
        d = Path(p).parents[1].name
        if d not in newest or os.path.getmtime(p) > os.path.getmtime(newest[d]):
            newest[d] = p
    for p in sorted(newest.values()):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k.startswith("diner::") or "points_mlp" in k:
                per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        agg = defaultdict(lambda: defaultdict(list))
        for (k, _), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
        for k, cs in agg.items():
            res.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
    if res:
        json.dump(res, open(out / f"pmc_{tag}.json", "w"), indent=1, sort_keys=True)
        print("wrote", out / f"pmc_{tag}.json")
    dom = [k for k in res if "points_mlp" in k and "FETCH_SIZE" in res[k] and "WRITE_SIZE" in res[k]]
    if dom:
        import bench
        k = max(dom, key=lambda n: res[n]["FETCH_SIZE"])
        byts = (2 * res[k]["FETCH_SIZE"] + res[k]["WRITE_SIZE"]) * 1024
        tj = ROOT / "profiles" / "traffic.json"
        t = json.loads(tj.read_text()) if tj.exists() else {}
        t = {kk: vv for kk, vv in t.items() if isinstance(vv, dict)}
        t[f"{a.config}:{a.precision}:{a.rays_per_call}"] = {
            "bytes_per_launch": byts, "kernel": k, "kernel_src_sha16": bench.kernel_source_digest(),
            "source": f"profiles/{a.round}/pmc_{tag}.json (separate rocprofv3 --pmc passes of `bench.py --config {a.config}`; (2*FETCH_SIZE + WRITE_SIZE) KiB; "
                      "L2-miss bytes: Infinity-Cache hits included)"}
        # the two small kernels of the north_star's "sampling + integration" clause, from the same passes: measured L2-miss bytes and,
        # for the sampler (VALU-bound: two erff and NV projections per candidate), the VALU-busy fraction
        # (SQ_ACTIVE_INST_VALU counts quad-cycles: x4 / 1024 SIMDs, over GRBM_GUI_ACTIVE / 8 XCDs -- MI355X_MICROARCH.md)
        small = {}
        for name, key in (("sampler", "sampler_kernel"), ("composite", "composite_kernel")):
            ks_ = [kk for kk in res if key in kk and "FETCH_SIZE" in res[kk] and "WRITE_SIZE" in res[kk]]
            if ks_:
                r_ = res[max(ks_, key=lambda n: res[n]["FETCH_SIZE"])]
                small[name] = {"bytes_per_launch": (2 * r_["FETCH_SIZE"] + r_["WRITE_SIZE"]) * 1024}
                if "SQ_ACTIVE_INST_VALU" in r_ and r_.get("GRBM_GUI_ACTIVE"):
                    small[name]["valu_busy_frac"] = r_["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (r_["GRBM_GUI_ACTIVE"] / 8)
        if small:
            t[f"{a.config}:{a.precision}:{a.rays_per_call}:sampling_integration"] = dict(
                small, kernel_src_sha16=bench.small_kernel_source_digest(),
                source=f"profiles/{a.round}/pmc_{tag}.json (same passes; (2*FETCH_SIZE + WRITE_SIZE) KiB per launch)")
        tj.write_text(json.dumps(t, indent=1, sort_keys=True))
        print("traffic", k, f"{byts / 1e9:.1f} GB per launch", {n: f"{v['bytes_per_launch'] / 1e9:.3f} GB" for n, v in small.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in res[k] and "GRBM_GUI_ACTIVE" in res[k]:
            busy = res[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (res[k]["GRBM_GUI_ACTIVE"] / 8)
            print(f"MFMA busy: {100 * busy:.1f} % of active cycles (per SIMD)")


if __name__ == "__main__":
    main()
