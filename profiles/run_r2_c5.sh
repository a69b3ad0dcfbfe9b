#!/bin/bash
# Round-2 c5 (regression NUTS, 1024 chains x 1e5 rows): tree statistics, then kernel-trace stats of the bench.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_c5
mkdir -p $O
python3 $R/tools/debug/c5_trees.py 1024 300 > $O/trees.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c5 -- python3 $R/bench.py --config c5 --steps 50 --warmup 200 --no-cpu-baseline > $O/stats.log 2>&1
tail -3 $O/stats.log
head -12 $O/stats/c5_kernel_stats.csv | cut -c1-220
