#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of profiles/run_r2_diag.sh into one JSON: per kernel the launch count,
average duration (kernel-trace stats) and the PMC counters averaged per launch.  FETCH_SIZE /
WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE tallies 128-byte requests at 64
bytes, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, sys

src = sys.argv[1]
out = collections.defaultdict(dict)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        out[k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]),
                      pct=float(r["Percentage"]))
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
        if "avg_ns" in d:
            d["hbm_GBs"] = d["hbm_bytes_per_launch"] / d["avg_ns"]
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("total_ns", 0))[:12]:
    print(k[:90], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in d.items()})

# per-transition HBM bytes of the two secondary workloads of bench.py -> profiles/r2/diag_pmc_summary.json
def find(sub):
    ks = [k for k in out if sub in k]
    return out[ks[0]] if ks else None

dm, nw, hw = find("k_draw_momentum"), find("k_nuts_wide"), find("k_hmc_wide")
summ = {"note": "rocprofv3 --pmc, separate passes (FETCH_SIZE; WRITE_SIZE; SQ groups) of tools/diag_run.py = bench.py's "
                "secondary workloads; FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B); averages per launch; "
                "one transition = one k_draw_momentum + one main-kernel launch"}
for name, main in (("nuts", nw), ("hmc", hw)):
    if main and dm and "hbm_bytes_per_launch" in main and "hbm_bytes_per_launch" in dm:
        summ[name] = {"hbm_bytes_per_transition": main["hbm_bytes_per_launch"] + dm["hbm_bytes_per_launch"],
                      "main_kernel": main, "k_draw_momentum": dm}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r2")
os.makedirs(dst, exist_ok=True)
json.dump(summ, open(os.path.join(src, "diag_pmc_summary.json"), "w"), indent=1)
