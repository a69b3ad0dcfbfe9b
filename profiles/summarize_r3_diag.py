#!/usr/bin/env python3
"""profiles/run_r3_diag.sh -> diag_pmc_summary.json: per kernel of bench.py's secondary workloads (D = 1e4 diagonal mass,
4096 chains) the launch count and average duration (kernel-trace stats) and the PMC counters averaged per launch, and the
HBM bytes per TRANSITION: NUTS = one k_draw_momentum + one k_nuts_wide launch; HMC (round 3) = one k_draw_momentum + one
k_hmc_wide launch per engine call of bench.HMC_PER_CALL transitions.  FETCH_SIZE / WRITE_SIZE come in KB; FETCH_SIZE is
doubled (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import HMC_PER_CALL


def fold(src):
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"].split("(")[0]
            out[k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, cn), v in agg.items():
            out[k][cn] = sum(v) / len(v)
            out[k].setdefault("pmc_launches", len(v))
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
            d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in d and "avg_ns" in d:
            d["clock_GHz_grbm"] = d["GRBM_GUI_ACTIVE"] / 8 / d["avg_ns"]
    return out


def find(out, sub):
    ks = [k for k in out if sub in k]
    return out[ks[0]] if ks else None


summ = {"note": "rocprofv3 --pmc, separate passes of tools/diag_run.py {nuts,hmc} = bench.py's secondary workloads; FETCH_SIZE doubled "
                "(gfx950); averages per launch; NUTS transition = k_draw_momentum + k_nuts_wide; HMC: one k_draw_momentum + one "
                f"k_hmc_wide launch per engine call of {HMC_PER_CALL} transitions (the position stays on chip in between)"}
for name, key, per in (("nuts", "k_nuts_wide", 1), ("hmc", "k_hmc_wide", HMC_PER_CALL)):
    out = fold(os.path.join(root, f"r3_diag_{name}"))
    main, dm = find(out, key), find(out, "k_draw_momentum")
    if main and dm and "hbm_bytes_per_launch" in main and "hbm_bytes_per_launch" in dm:
        summ[name] = {"hbm_bytes_per_transition": (main["hbm_bytes_per_launch"] + dm["hbm_bytes_per_launch"]) / per,
                      "transitions_per_launch": per, "kernel_ms_per_transition": (main["avg_ns"] + dm["avg_ns"]) / per / 1e6,
                      "main_kernel": main, "k_draw_momentum": dm}
json.dump(summ, open(os.path.join(root, "r3_diag_pmc_summary.json"), "w"), indent=1)
for k in ("nuts", "hmc"):
    if k in summ:
        s = summ[k]
        print(k, "bytes/transition %.3e" % s["hbm_bytes_per_transition"], "kernel ms/transition %.3f" % s["kernel_ms_per_transition"],
              "main VALU/wave %.0f" % s["main_kernel"].get("valu_per_wave", 0), "wait %.2f" % s["main_kernel"].get("wait_any_frac", 0),
              "clock %.2f" % s["main_kernel"].get("clock_GHz_grbm", 0))
