#!/bin/bash
# Round 3: counters of the small-dense NUTS kernel (tools/debug/small_dense.py 50 4096: k_nuts_resident<64,1,true,3>,
# sample(100) of 4096 chains, dense metric and dense target at D = 50).  Separate --pmc passes, program after `--`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_small_dense_pmc
mkdir -p $O
CMD="python3 $R/tools/debug/small_dense.py 50 4096 2"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o s -- $CMD > $O/pmc_$tag.log 2>&1 < /dev/null
done
python3 - $O <<'PY'
import csv, glob, json, os, sys
src = sys.argv[1]
k = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_nuts_resident" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)  # the timed sample(100) launch
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            k["kernel"] = r["Kernel_Name"][:60]
wc = k.get("SQ_WAVE_CYCLES", 0.0)
d = {"valu_per_wave": k.get("SQ_INSTS_VALU", 0) / max(k.get("SQ_WAVES", 1), 1),
     "lds_per_wave": k.get("SQ_INSTS_LDS", 0) / max(k.get("SQ_WAVES", 1), 1),
     "salu_per_wave": k.get("SQ_INSTS_SALU", 0) / max(k.get("SQ_WAVES", 1), 1),
     "f64_fma_add_mul_trans_share_of_valu": (k.get("SQ_INSTS_VALU_FMA_F64", 0) + k.get("SQ_INSTS_VALU_ADD_F64", 0) +
                                             k.get("SQ_INSTS_VALU_MUL_F64", 0) + k.get("SQ_INSTS_VALU_TRANS_F64", 0)) / max(k.get("SQ_INSTS_VALU", 1), 1),
     "active_valu_frac_of_wave_cycles": k.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None,
     "active_lds_frac_of_wave_cycles": k.get("SQ_ACTIVE_INST_LDS", 0) / wc if wc else None,
     "wait_inst_any_frac_of_wave_cycles": k.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
     "wait_any_frac_of_wave_cycles": k.get("SQ_WAIT_ANY", 0) / wc if wc else None,
     "hbm_bytes": k.get("FETCH_SIZE", 0) * 1024 * 2 + k.get("WRITE_SIZE", 0) * 1024,
     "grbm_gui_active": k.get("GRBM_GUI_ACTIVE", 0)}
json.dump({"note": "rocprofv3 --pmc passes of tools/debug/small_dense.py 50 4096 2 (profiles/run_r3_small_dense_pmc.sh); last "
                   "k_nuts_resident dispatch = sample(100) of 4096 chains, D = 50, dense metric and dense target; FETCH_SIZE "
                   "doubled (gfx950); SQ cycle counters in quad-cycles", "raw": k, "derived": d},
          open(os.path.join(src, "small_dense_pmc_summary.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
PY
