#!/bin/bash
# round 4: block-resident dense kernels -- parity / bitwise tests, then timings against the lock-step path
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4b
timeout 1200 python -m pytest tests/test_gpu_block_dense.py -x -q 2>&1 | tail -25
timeout 600 python tools/debug/mid_dense.py 2>&1 | tail -30
