#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_block_dense.py -x -q 2>&1 | tail -5
for D in 200; do
  AEHMC_AMD_LIB=$PWD/aehmc_amd/libaehmc_hip_timing.so timeout 300 python tools/debug/block_phases.py $D 4096 10 2>&1 | grep -v amdgpu.ids
done
for D in 100 200 256 500; do timeout 300 python tools/debug/mid_dense.py $D 4096 10 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/debug/mid_dense.py 200 4096 5 32 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/debug/mid_dense.py 500 4096 5 32 2>&1 | grep -v amdgpu.ids
./tools/bin/blk_gemm_bench 200 4096 200 | grep -v amdgpu; ./tools/bin/blk_gemm_bench 512 4096 200 | grep -v amdgpu
timeout 900 bash profiles/run_r4.sh c1 c2 2>&1 | tail -12
