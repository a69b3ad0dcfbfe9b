#!/bin/bash
# Generic rocprofv3 recipe (run through gpurun from the repo root): kernel-trace stats, then separate
# --pmc passes (no trace domains mixed in) of one command.  usage: pmc_run.sh <tag> <script path relative to the repo> [args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
S=$R/$1
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $S "$@" > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o run -- python3 $S "$@" > $O/pmc_$tag.log 2>&1
done
python3 $R/profiles/summarize_r2_diag.py $O
