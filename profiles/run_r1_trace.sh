cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
rm -rf $O/prof_c3b
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3b -o c3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_c3b.log 2>&1
tail -1 $O/prof_c3b.log | cut -c1-300
