#!/bin/bash
# k_step_linear<15,true> and GEMM average durations of a 2-step c3 run, per library variant (AEHMC_AMD_LIB)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  O=$R/gpurun_out/stage_$v
  rm -rf $O; mkdir -p $O
  export AEHMC_AMD_LIB=$R/aehmc_amd/libaehmc_hip$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1
  echo "lib$v: $(grep -E 'k_step_linear<15|streamk_kernel<true, 8' $O/c3_kernel_stats.csv | awk -F'","|",' '{print substr($1,1,45), $4}' | tr '\n' ' ') $(tail -1 $O/log.txt | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms", d["ms_per_step"])')"
  find $O -name "*kernel_trace.csv" -delete
done
