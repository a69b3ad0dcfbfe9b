#!/bin/bash
# Round-6 counters (round 5's script), one directory per workload under gpurun_out/r6_<tag>: kernel-trace stats, then SEPARATE --pmc passes
# (no trace domains mixed in), the program directly after `--`.  Run through gpurun from the repo root with the FINAL
# libaehmc_hip.so; profiles/summarize_r6.py folds the outputs into profiles/r6/*_pmc_summary.json, each stamped with the
# sha256 of the library it measured (bench.py drops counter-derived figures whose stamp is not the loaded library's).
# usage: run_r6.sh <tag> [<tag> ...]   tags: c1 c2 c2_fc c3 c5 diag_nuts diag_hmc diag_hmc_fc mid200 mid100 pc200
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS|SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES|SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64|GRBM_GUI_ACTIVE"
for tag in "$@"; do
  groups="FETCH_SIZE|WRITE_SIZE|$SQ"
  case $tag in
    c1) CMD="python3 $R/tools/c1_run.py 20"; groups="$groups|SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" ;;
    c2) CMD="python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c2_fc) CMD="python3 $R/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --fp-contract" ;;
    c3) CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
        groups="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE" ;;
    c5) CMD="python3 $R/tools/c5_run.py 1024 1000 1000"; groups="$groups|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ;;
    diag_nuts) CMD="python3 $R/tools/diag_run.py nuts 2" ;;
    diag_hmc) CMD="python3 $R/tools/diag_run.py hmc 2" ;;
    diag_hmc_fc) CMD="python3 $R/tools/diag_run.py hmc 2 10000 4096 1" ;;
    mid200) CMD="python3 $R/tools/debug/mid_dense.py 200 4096 10"; groups="$groups|SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR|SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" ;;
    mid100) CMD="python3 $R/tools/debug/mid_dense.py 100 4096 10"; groups="$groups|SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR|SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" ;;
    pc200) CMD="python3 $R/tools/debug/pc_dense_time.py 200 4096 10"; groups="FETCH_SIZE|WRITE_SIZE|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS|SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES|GRBM_GUI_ACTIVE|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ;;
    *) echo "unknown tag $tag"; continue ;;
  esac
  O=$R/gpurun_out/r6_$tag
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $CMD > $O/stats.log 2>&1
  IFS='|' read -ra GR <<< "$groups"
  for grp in "${GR[@]}"; do
    t=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$t -o run -- $CMD > $O/pmc_$t.log 2>&1
  done
  find $O -name "*kernel_trace.csv" -size +20M -delete
  echo "$CMD" > $O/command.txt
  sha256sum $R/aehmc_amd/libaehmc_hip.so | cut -d' ' -f1 > $O/lib_sha256.txt
done
python3 $R/profiles/summarize_r6.py $R/gpurun_out "$@"
