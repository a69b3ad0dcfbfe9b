#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun; outputs land in gpurun_out/).
# 1) kernel-trace + stats of the default bench command (c3) and of the c2 bench
# 2) separate PMC passes (FETCH_SIZE, WRITE_SIZE) on a 1-transition c3 run and a short c2 run
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 $R/bench.py --steps 3 --warmup 1 > $O/prof_c3_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 $R/bench.py --config c2 --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_c2_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c3 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c3 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c2 -o c2 -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_c2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c2 -o c2 -- python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_write_c2.log 2>&1
ls -la $O/prof_c3 $O/pmc_fetch | head -30
# keep only the small summaries (traces can be large)
find $O -name "*kernel_trace.csv" -size +20M -delete
tail -2 $O/prof_c3_bench.log
