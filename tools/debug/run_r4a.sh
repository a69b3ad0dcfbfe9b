#!/bin/bash
# round 4, first GPU call: fp_contract tests + the default bench line (with both modes in `secondary`)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_fp_contract.py tests/test_abi.py -x -q 2>&1 | tail -15
timeout 900 python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4a/bench_default.json") if l.startswith("{")][-1])
print("c3", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("counters_dropped"))
for s in d["secondary"]:
    print(s.get("config"), s.get("value"), (s.get("roofline") or {}).get("frac"), s.get("error"))
PY
