#!/bin/bash
# Round-3 counters for the headline config c3: separate --pmc passes (no trace domains mixed in) of a 1-transition run of
# the default bench command, program directly after `--`; kernel-trace stats of the same command for the durations.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_c3
mkdir -p $O
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c3 -- $CMD > $O/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o c3 -- $CMD > $O/pmc_$tag.log 2>&1
done
find $O -name "*kernel_trace.csv" -size +20M -delete
python3 $R/profiles/summarize_r3_c3.py $O
