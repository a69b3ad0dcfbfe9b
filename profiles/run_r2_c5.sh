#!/bin/bash
# Round-2 c5 (regression NUTS + window adaptation, 1024 chains x 1e5 rows): bench lines, kernel-trace
# stats of the bench, the row-sweep microbenchmark, the per-phase breakdown of k_nuts_linreg.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_c5
mkdir -p $O
python3 $R/bench.py --config c5 --steps 10 --warmup 1000 > $O/bench_c5_10.json 2> $O/bench_c5_10.err
python3 $R/bench.py --config c5 --steps 100 --warmup 1000 --no-cpu-baseline > $O/bench_c5_100.json 2> $O/bench_c5_100.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c5 -- python3 $R/bench.py --config c5 --steps 100 --warmup 1000 --no-cpu-baseline > $O/stats.log 2>&1
head -8 $O/stats/c5_kernel_stats.csv | cut -c1-200
$R/tools/bin/lr_stream_bench > $O/row_sweep_microbench.txt 2>&1
AEHMC_AMD_LIB=$R/aehmc_amd/libaehmc_hip_timing.so python3 $R/tools/debug/linreg_phases.py 1024 100 > $O/linreg_phases.txt 2>&1
python3 $R/tools/debug/c5_trees.py 1024 300 > $O/trees.txt 2>&1
cat $O/bench_c5_10.json $O/bench_c5_100.json $O/linreg_phases.txt
