#!/bin/bash
# round 4: block-resident dense kernels, register version -- tests, phases, timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_block_dense.py -x -q 2>&1 | tail -15
for D in 100 200; do
  AEHMC_AMD_LIB=$PWD/aehmc_amd/libaehmc_hip_timing.so timeout 300 python tools/debug/block_phases.py $D 4096 10 2>&1 | grep -v amdgpu.ids
done
AEHMC_AMD_LIB=$PWD/aehmc_amd/libaehmc_hip_timing.so timeout 300 python tools/debug/block_phases.py 512 4096 10 2>&1 | grep -v amdgpu.ids
for D in 100 200 256 500; do timeout 300 python tools/debug/mid_dense.py $D 4096 10 2>&1 | grep -v amdgpu.ids; done
for D in 100 200; do BLOCK_DENSE=2 timeout 300 python tools/debug/mid_dense.py $D 4096 10 2>&1 | grep -v amdgpu.ids; done
for D in 100 200; do BLOCK_DENSE=0 timeout 300 python tools/debug/mid_dense.py $D 4096 10 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/debug/mid_dense.py 200 4096 5 32 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/debug/mid_dense.py 200 16384 5 2>&1 | grep -v amdgpu.ids
