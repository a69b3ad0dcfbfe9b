#!/bin/bash
# Round-3 bench lines (the driver's default command and the other single-GPU configs) and the kernel-trace stats of
# the default command at 3 steps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_final
mkdir -p $O
cd $R
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c1 > $O/bench_c1.json 2> $O/bench_c1.err
for f in c3 c2 c5 c1; do tail -1 $O/bench_$f.json | cut -c1-300; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
head -8 $O/stats/c3_kernel_stats.csv | cut -c1-160
