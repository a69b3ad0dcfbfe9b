#!/bin/bash
# copies what profiles/run_r6.sh left under gpurun_out/ into profiles/r6/ (run from the repo root, after the gpurun call)
set -e
for t in c1 c2 c2_fc c3 c5 diag_nuts diag_hmc diag_hmc_fc; do cp gpurun_out/r6_$t/stats/run_kernel_stats.csv profiles/r6/${t}_kernel_stats.csv; done
for t in mid100 mid200 pc200; do cp gpurun_out/r6_$t/stats/run_kernel_stats.csv profiles/r6/dense/${t}_kernel_stats.csv; cp gpurun_out/r6_summaries/${t}_pmc_summary.json profiles/r6/dense/; done
for t in c1 c2 c2_fc c3 c5 diag; do cp gpurun_out/r6_summaries/${t}_pmc_summary.json profiles/r6/; done
grep -h lib_sha256 profiles/r6/*_pmc_summary.json profiles/r6/dense/*_pmc_summary.json | sort | uniq -c
