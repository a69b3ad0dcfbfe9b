#!/bin/bash
# SQ / GRBM counters of the fp64 GEMM at M=4096, N=K=1e4 (tools/gemm_once.py M [streamk mode]); separate
# passes, counters only (no sys/hip traces).  Run through gpurun from the repo root; prints per-launch averages.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_gemm_sq
MODE=${1:-2}
rm -rf $O
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $set | cut -c1-12 | tr " " _)
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$n -o g -- python3 $R/tools/gemm_once.py 4096 $MODE > /dev/null 2>&1
done
find $O -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob, collections
tot = {}
for f in sorted(glob.glob("$O/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
for k, v in tot.items():
    print(f"{k:28s} {v:.4g}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
    # GRBM_GUI_ACTIVE sums 8 XCDs; MFMA busy cycles sum 1024 SIMDs
    print("MFMA pipe busy = %.3f of the kernel's cycles" % ((tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (tot["GRBM_GUI_ACTIVE"] / 8)))
PY
