#!/bin/bash
# Round-3 counters for config c5 (k_nuts_linreg: 1024 chains x 1e5 rows, sample(1000) after a 1000-step warm-up = the
# bench's default launch):
# kernel-trace stats, then SEPARATE --pmc passes (no trace domains mixed in), program directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_c5
mkdir -p $O
CMD="python3 $R/tools/c5_run.py 1024 1000 1000"
$CMD > $O/run.json 2> $O/run.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c5 -- $CMD > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o c5 -- $CMD > $O/pmc_$tag.log 2>&1
done
python3 $R/profiles/summarize_r3_c5.py $O
