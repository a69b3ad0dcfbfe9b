cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
rm -rf $O/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $R/tools/c5_bench.py 1024 300 20 > $O/prof_c5.log 2>&1
grep -E "warm-up|sampling" $O/prof_c5.log
head -8 $O/prof_c5/c5_kernel_stats.csv | cut -c1-200
