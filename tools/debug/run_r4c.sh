#!/bin/bash
# round 4: adaptation building blocks + phase breakdown of the block-resident dense kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_building_blocks.py tests/test_gpu_block_dense.py -x -q 2>&1 | tail -15
for D in 100 200 512; do
  AEHMC_AMD_LIB=$PWD/aehmc_amd/libaehmc_hip_timing.so timeout 300 python tools/debug/block_phases.py $D 4096 10 2>&1 | grep -v amdgpu.ids
done
for D in 100 200 500; do timeout 300 python tools/debug/mid_dense.py $D 4096 10 2>&1 | grep -v amdgpu.ids; done
timeout 300 python tools/debug/mid_dense.py 200 4096 5 32 2>&1 | grep -v amdgpu.ids
