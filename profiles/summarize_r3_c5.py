#!/usr/bin/env python3
"""profiles/run_r3_c5.sh -> c5_pmc_summary.json: counters of the LAST k_nuts_linreg dispatch of tools/c5_run.py (the
sample() launch of 1024 chains x 1e5 rows after the warm-up launch) and the derived figures bench.py --config c5
reports.  FETCH_SIZE / WRITE_SIZE come in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B,
MI355X_MICROARCH.md, HBM section); SQ_*CYCLES / SQ_WAIT* / SQ_ACTIVE* count quad-cycles."""
import csv, glob, json, os, sys

src = sys.argv[1]
KEY = "k_nuts_linreg"
run = json.loads(open(os.path.join(src, "run.json")).read().strip().splitlines()[-1])
k = {}
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if KEY in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-1]
    k["launch_ns"] = int(last["End_Timestamp"]) - int(last["Start_Timestamp"])
    k["warmup_launch_ns"] = int(rows[0]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    k["vgpr"], k["sgpr"], k["lds_bytes"] = last.get("VGPR_Count"), last.get("SGPR_Count"), last.get("LDS_Block_Size")
    k["grid"], k["workgroup"] = last.get("Grid_Size"), last.get("Workgroup_Size")
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if KEY in r["Kernel_Name"]]
    if not rows:
        continue
    last_id = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last_id:
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
t = k["launch_ns"] * 1e-9
nl, N, C = run["leapfrogs"], run["rows"], run["chains"]
rounds_lb = nl / 4.0  # a sweep serves up to four chains: lower bound on the sweeps of the launch
d = {"leapfrogs": nl, "launch_ms": t * 1e3, "leapfrogs_per_s_kernel": nl / t,
     "fp64_fma_tflops_algorithmic": 6.0 * N * nl / t / 1e12, "frac_of_fp64_vector_peak_78.6": 6.0 * N * nl / t / 78.6e12}
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    d["hbm_bytes_per_launch"] = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024
    d["hbm_GBs"] = d["hbm_bytes_per_launch"] / t / 1e9
if "SQ_INSTS_VALU_FMA_F64" in k:
    d["fma_f64_wave_instructions"] = k["SQ_INSTS_VALU_FMA_F64"]
    d["fma_f64_per_row_chain_leapfrog"] = k["SQ_INSTS_VALU_FMA_F64"] * 64 / (N * nl)
    d["fma_f64_tflops_counted"] = k["SQ_INSTS_VALU_FMA_F64"] * 64 * 2 / t / 1e12
if "SQ_INSTS_VALU" in k and "SQ_WAVES" in k:
    d["valu_instructions_per_wave"] = k["SQ_INSTS_VALU"] / k["SQ_WAVES"]
    d["fma_share_of_valu"] = k.get("SQ_INSTS_VALU_FMA_F64", 0) / k["SQ_INSTS_VALU"]
if "SQ_WAVE_CYCLES" in k:
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
        if n in k:
            d[n.lower() + "_frac_of_wave_cycles"] = k[n] / k["SQ_WAVE_CYCLES"]
if "SQ_ACTIVE_INST_VALU" in k:
    d["valu_busy_fraction_of_kernel_time_at_2.4GHz"] = k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e9 * t)
if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k:
    d["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1)
if "TCC_REQ_sum" in k:
    d["l2_requests"] = k["TCC_REQ_sum"]
    d["l2_bytes_at_128B_per_request"] = k["TCC_REQ_sum"] * 128
    d["l2_TBs_at_128B_per_request"] = k["TCC_REQ_sum"] * 128 / t / 1e12
    d["l2_bytes_algorithmic"] = 16.0 * (N - 10176) * rounds_lb
if "GRBM_GUI_ACTIVE" in k:
    d["clock_GHz_grbm"] = k["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
out = {"note": "rocprofv3, separate --pmc passes of `tools/c5_run.py` (profiles/run_r3_c5.sh; chains, warm-up steps and transitions in `run`); values of the LAST "
               "k_nuts_linreg dispatch = the sample() launch after the warm-up launch; FETCH_SIZE "
               "doubled; SQ cycle counters in quad-cycles",
       "run": run, "per_launch": k, "derived": d}
json.dump(out, open(os.path.join(src, "c5_pmc_summary.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
