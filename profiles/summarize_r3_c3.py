#!/usr/bin/env python3
"""profiles/run_r3_c3.sh -> c3_pmc_summary.json: per kernel of one c3 transition the launch count, average duration
(kernel-trace stats) and PMC counters averaged per launch; `gemm_summary` for the dominant kernel in the layout bench.py
reads.  FETCH_SIZE / WRITE_SIZE come in KB, FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B:
MI355X_MICROARCH.md, HBM section); TCC_* count 128-byte L2 requests summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles, SQ_BUSY_CYCLES is summed over the shader engines."""
import collections, csv, glob, json, os, sys

src = sys.argv[1]
out = collections.defaultdict(dict)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        out[k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]), pct=float(r["Percentage"]))
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
    if "GRBM_GUI_ACTIVE" in d and "avg_ns" in d:
        d["clock_GHz_grbm"] = d["GRBM_GUI_ACTIVE"] / 8 / d["avg_ns"]
name = next(k for k in out if "gemm_nt_f64_streamk_kernel<true, 8" in k)
g = out[name]
D = 10_000
summ = {"kernel": name, "launches": g.get("pmc_launches"),
        "traffic_bytes_per_launch_avg": g["hbm_bytes_per_launch"],
        "fetch_bytes_avg_x2_gfx950": 2 * g["FETCH_SIZE"] * 1024, "write_bytes_avg": g["WRITE_SIZE"] * 1024,
        "avg_launch_ms_rocprof": g["avg_ns"] / 1e6, "clock_GHz_grbm": g.get("clock_GHz_grbm")}
if "TCC_REQ_sum" in g:
    summ.update(l2_requests_per_launch=g["TCC_REQ_sum"], l2_hits=g["TCC_HIT_sum"], l2_misses=g["TCC_MISS_sum"],
                l2_hit_rate=g["TCC_HIT_sum"] / max(g["TCC_HIT_sum"] + g["TCC_MISS_sum"], 1),
                l2_request_bytes_per_launch=g["TCC_REQ_sum"] * 128)
if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
    # MFMA pipe busy cycles summed over the 1024 SIMDs / (SIMDs x the launch's shader cycles)
    summ["mfma_busy_fraction"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g["GRBM_GUI_ACTIVE"] / 8)
doc = {"note": "rocprofv3, separate --pmc passes of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary` "
               "(profiles/run_r3_c3.sh); averages over the GEMM launches of one c3 transition (live rows shrink along the "
               "transition); FETCH_SIZE doubled (gfx950)",
       "gemm_summary": summ,
       "other_kernels": {k: {a: b for a, b in d.items() if a in ("calls", "avg_ns", "pct", "hbm_bytes_per_launch", "hbm_GBs", "clock_GHz_grbm")}
                         for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("total_ns", 0))[:10] if k != name}}
json.dump(doc, open(os.path.join(src, "c3_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
for k, d in list(doc["other_kernels"].items())[:3]:
    print(k[:70], d)
