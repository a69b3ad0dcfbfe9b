#!/bin/bash
# Round-2 profiling of the diagonal-mass kernels at D = 1e4, 4096 chains (run through gpurun from the
# repo root).  Kernel-trace stats, then SEPARATE --pmc passes (no trace domains mixed in):
# HBM-side bytes (FETCH_SIZE, WRITE_SIZE) and SQ instruction / cycle counters.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_diag
mkdir -p $O
W=${1:-both}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o diag -- python3 $R/tools/diag_run.py $W 3 > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o diag -- python3 $R/tools/diag_run.py $W 2 > $O/pmc_$tag.log 2>&1
done
python3 $R/profiles/summarize_r2_diag.py $O
