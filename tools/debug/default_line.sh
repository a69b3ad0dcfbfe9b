#!/bin/bash
# wall time of the driver's default command and a digest of its line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
S=$(date +%s.%N)
python bench.py > gpurun_out/r3_default_line.json 2> gpurun_out/r3_default_line.err
E=$(date +%s.%N)
echo "wall seconds: $(echo "$E - $S" | bc)"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_default_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_source"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][-40:], d["cpu_baseline"]["single_thread"]["value"])
for s in d["secondary"]:
    r = s.get("roofline") or {}
    print(s["config"], "%.4g" % s["value"], r.get("bound"), r.get("frac"), (r.get("hbm") or {}).get("frac"), r.get("traffic_source"))
PY
