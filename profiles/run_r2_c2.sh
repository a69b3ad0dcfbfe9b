#!/bin/bash
# Round-2 counters for config c2 (k_hmc_fused<2, iso>, 100 transitions of 4096 chains per launch):
# kernel-trace stats, then separate --pmc passes (HBM bytes, SQ instruction / cycle counters).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_c2
mkdir -p $O
CMD="python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- $CMD > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o c2 -- $CMD > $O/pmc_$tag.log 2>&1
done
python3 $R/profiles/summarize_r2_diag.py $O | grep -E "k_hmc_fused"
python3 $R/profiles/summarize_r2_c2.py $O
