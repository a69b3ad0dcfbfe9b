#!/bin/bash
# Round-2 evidence for the headline config c3: kernel-trace stats of the default bench command, then
# separate --pmc passes on a 1-transition run (HBM-side bytes; L2 hit / miss; MFMA busy).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_c3
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o c3 -- $CMD > $O/pmc_$tag.log 2>&1
done
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 $R/profiles/summarize_r2_diag.py $O | grep -E "gemm|k_step_linear|k_compact" | cut -c1-1200
