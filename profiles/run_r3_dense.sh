#!/bin/bash
# Round 3: kernel-trace stats of the small / mid-size dense workloads (tools/debug/*.py) outside the benchmarked configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_dense
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, script, args...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python3 "$@" > $O/$name.log 2>&1 < /dev/null
  grep -h "leapfrog/s\|warm-up\|^M=" $O/$name.log | tail -4
  [ -f $O/$name/s_kernel_stats.csv ] && cp $O/$name/s_kernel_stats.csv $O/${name}_kernel_stats.csv
  find $O/$name -name "*kernel_trace.csv" -delete
}
run small_dense_nuts_d50 $R/tools/debug/small_dense.py 50 4096 2
run small_dense_hmc_d50 $R/tools/debug/small_dense.py 50 4096 2 16
run mid_dense_nuts_d200 $R/tools/debug/mid_dense.py 200 4096 10
run mid_dense_nuts_d500 $R/tools/debug/mid_dense.py 500 4096 10
run full_adapt_d20 $R/tools/debug/full_adapt_time.py 20 4096 300 2
run gemm_mid_4096x200 $R/tools/debug/gemm_mid.py 4096x200
run gemm_mid_4096x500 $R/tools/debug/gemm_mid.py 4096x500
run gemm_mid_1024x1000 $R/tools/debug/gemm_mid.py 1024x1000
ls $O/*_kernel_stats.csv
