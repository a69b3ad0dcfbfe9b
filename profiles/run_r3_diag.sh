#!/bin/bash
# Round-3 profiling of the diagonal-mass kernels at D = 1e4, 4096 chains (bench.py's `secondary` workloads),
# NUTS and HMC in separate runs (both use k_draw_momentum, at different sizes).  Kernel-trace stats, then
# SEPARATE --pmc passes (no trace domains mixed in), program directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in nuts hmc; do
  O=$R/gpurun_out/r3_diag_$W
  mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o diag -- python3 $R/tools/diag_run.py $W 3 > $O/stats.log 2>&1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o diag -- python3 $R/tools/diag_run.py $W 2 > $O/pmc_$tag.log 2>&1
  done
  find $O -name "*kernel_trace.csv" -size +20M -delete
done
python3 $R/profiles/summarize_r3_diag.py $R/gpurun_out
