#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy of one translation unit of the library (hipcc remarks), one line per
kernel.  usage: resource_usage.py <tu.hip> [name-filter]"""
import re, subprocess, sys, os
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aehmc_amd", "csrc")
flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function -mllvm -disable-machine-licm".split()
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null",
                      sys.argv[1]], cwd=csrc, capture_output=True, text=True).stderr
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = {}
def flush():
    if cur and filt in cur.get("name", ""):
        dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("aehmc::", "")
        print(f"{dem:50s} VGPR {cur.get('VGPRs','?'):>3} AGPR {cur.get('AGPRs','?'):>3} SGPR {cur.get('TotalSGPRs','?'):>3} "
              f"scratch {cur.get('ScratchSize [bytes/lane]','?'):>4} occ {cur.get('Occupancy [waves/SIMD]','?')} LDS {cur.get('LDS Size [bytes/block]','?')}")
for l in out.split("\n"):
    m = re.search(r"remark: [^ ]+ +(?:Function Name|Name): (\S+)", l) or re.search(r"Function Name: (\S+)", l) or re.search(r" Name: (\S+)", l)
    if m:
        flush(); cur = {"name": m.group(1)}; continue
    m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", l)
    if m and cur: cur[m.group(1)] = m.group(2)
flush()
