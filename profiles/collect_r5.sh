#!/bin/bash
# copies what profiles/run_r5.sh left under gpurun_out/ into profiles/r5/ (run from the repo root, after the gpurun call)
set -e
for t in c1 c2 c2_fc c3 c5 diag_nuts diag_hmc diag_hmc_fc; do cp gpurun_out/r5_$t/stats/run_kernel_stats.csv profiles/r5/${t}_kernel_stats.csv; done
for t in mid100 mid200 pc200; do cp gpurun_out/r5_$t/stats/run_kernel_stats.csv profiles/r5/dense/${t}_kernel_stats.csv; cp gpurun_out/r5_summaries/${t}_pmc_summary.json profiles/r5/dense/; done
for t in c1 c2 c2_fc c3 c5 diag; do cp gpurun_out/r5_summaries/${t}_pmc_summary.json profiles/r5/; done
grep -h lib_sha256 profiles/r5/*_pmc_summary.json profiles/r5/dense/*_pmc_summary.json | sort | uniq -c
