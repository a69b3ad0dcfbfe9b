#!/bin/bash
# Regenerates everything under profiles/r1 except the MFMA peak probe (run through gpurun from the
# repo root; every step is bounded).  Afterwards, here:  python profiles/summarize_pmc.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
timeout 900 bash $R/profiles/run_r1_profiles.sh > $O/refresh_profiles.log 2>&1
cd $R
timeout 300 python bench.py > $O/c3_bench.json 2> $O/c3_bench.err
timeout 200 python bench.py --config c2 --steps 20 --warmup 3 > $O/c2_bench.json 2> $O/c2_bench.err
timeout 200 python bench.py --config c1 --steps 20 --warmup 2 > $O/c1_bench.json 2> $O/c1_bench.err
timeout 300 python bench.py --config c5 --steps 10 --warmup 1000 > $O/c5_bench.json 2> $O/c5_bench.err
{
  for cfg in "10000 4096" "4000 4096" "1000 4096" "100 32768" "100 4096" "2 65536"; do timeout 300 python tools/nuts_diag_bench.py $cfg 2>/dev/null | grep resident; done
  timeout 200 python tools/hmc_diag_bench.py 10000 4096 32 10 2>/dev/null | grep resident
  timeout 200 python tools/hmc_diag_bench.py 4096 4096 32 10 2>/dev/null | grep resident
  timeout 200 python tools/linreg_nuts_bench.py 1024 2e-4 2>/dev/null | grep resident
} > $O/diag_nuts_bench.txt
{
  timeout 300 python tools/gemm_time.py 4096 3500 2949 2900 2048 1100 2>/dev/null | grep streamk
  timeout 200 python tools/vendor_dgemm_time.py 4096 2949 2048 2>/dev/null | grep vendor
  timeout 100 python tools/rng_bench.py 2>/dev/null | grep -E "normals|bernoulli"
  timeout 60 ./tools/bin/rng_probe 2>/dev/null
} > $O/gemm_rng_timing.txt
grep -h '^{"metric"' $O/prof_c3_bench.log > $O/c3_bench_under_rocprof.json
grep -h '^{"metric"' $O/prof_c2_bench.log > $O/c2_bench_under_rocprof.json
tail -1 $O/c3_bench.json | cut -c1-200
