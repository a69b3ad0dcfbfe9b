#!/usr/bin/env python3
"""Per kernel of libaehmc_hip.so: the ScratchSize the compiler reserves AND the number of scratch instructions in its
ISA.  A reservation without instructions is the frame slot of SGPRs spilled to VGPR lanes (v_writelane / v_readlane):
the kernel never touches scratch memory.  usage: scratch_report.py [engine.s]   (compiles csrc/engine.hip to ISA when
no file is given: ~3 minutes)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = os.path.join(tempfile.gettempdir(), "aehmc_engine_full.s")
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -mllvm -disable-machine-licm".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", path, "engine.hip"],
                          cwd=os.path.join(ROOT, "aehmc_amd", "csrc"), stderr=subprocess.DEVNULL)
name, n_scr, n_lane, rows = None, 0, 0, []
for line in open(path):
    m = re.match(r"^(_Z\w+):\s", line)
    if m:
        name, n_scr, n_lane = m.group(1), 0, 0
    elif "scratch_load" in line or "scratch_store" in line:
        n_scr += 1
    elif "v_writelane_b32" in line or "v_readlane_b32" in line:
        n_lane += 1
    else:
        m = re.match(r"^; ScratchSize: (\d+)", line)
        if m and name:
            rows.append((name, int(m.group(1)), n_scr, n_lane))
            name = None
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{len(rows)} kernels; {sum(1 for r in rows if r[1])} reserve scratch; {sum(1 for r in rows if r[2])} execute scratch instructions")
print("reserved B | scratch instrs | lane spills | kernel")
for (n, sz, ns, nl), d in zip(rows, names):
    if sz or ns:
        print(f"{sz:10d} | {ns:14d} | {nl:11d} | {d.split('(')[0]}")
