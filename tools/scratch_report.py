#!/usr/bin/env python3
"""Per kernel of libaehmc_hip.so: the ScratchSize the compiler reserves AND the number of scratch instructions in its
ISA, by loop depth (0: straight-line prologue / epilogue, 1: the outermost loop -- the transition loop of the
many-transition kernels, the round loop of k_nuts_block_roll / the leapfrog loop of k_nuts_wide --, 2+: inner loops, i.e.
per leapfrog in the kernels that run several transitions).  A reservation without instructions is the frame slot of
SGPRs spilled to VGPR lanes (v_writelane / v_readlane): the kernel never touches scratch memory.  usage: scratch_report.py [unit.s ...]   (compiles every csrc/*.hip to ISA
when no file is given: ~2 minutes)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    paths = sys.argv[1:]
else:  # every translation unit of the library (round 5: engine.hip + one tu_*.hip per kernel family), compiled in parallel
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "aehmc_amd", "csrc")
    flags = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -mllvm -disable-machine-licm".split()
    units = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    paths = [os.path.join(tempfile.gettempdir(), "aehmc_" + u[:-4] + ".s") for u in units]

    def compile_unit(k):
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DAEHMC_GPU_ARCH=\"gfx950\"", "-S", "--cuda-device-only", "-o", paths[k],
                               units[k]], cwd=csrc, stderr=subprocess.DEVNULL)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_unit, range(len(units))))
name, n_scr, n_lane, rows, depth, by_depth = None, 0, 0, [], 0, {}
import itertools
seen = set()
for line in itertools.chain.from_iterable(open(pth) for pth in paths):
    m = re.match(r"^(_Z\w+):\s", line)
    if m:
        name, n_scr, n_lane, depth, by_depth = m.group(1), 0, 0, 0, {}
    elif re.match(r"^\.LBB", line):
        d = re.findall(r"Depth=(\d+)", line)
        depth = int(d[0]) if d else 0
    elif "scratch_load" in line or "scratch_store" in line:
        n_scr += 1
        by_depth[depth] = by_depth.get(depth, 0) + 1
    elif "v_writelane_b32" in line or "v_readlane_b32" in line:
        n_lane += 1
    else:
        m = re.match(r"^; ScratchSize: (\d+)", line)
        if m and name:
            if name not in seen:  # (helper kernels of engine.cuh are compiled into every unit that includes it)
                seen.add(name)
                rows.append((name, int(m.group(1)), n_scr, n_lane, dict(by_depth)))
            name = None
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{len(rows)} kernels; {sum(1 for r in rows if r[1])} reserve scratch; {sum(1 for r in rows if r[2])} execute scratch instructions")
print("reserved B | scratch instrs (depth 0 / 1 / 2+) | lane spills | kernel")
for (n, sz, ns, nl, bd), d in zip(rows, names):
    if sz or ns:
        deep = sum(v for k, v in bd.items() if k >= 2)
        print(f"{sz:10d} | {ns:6d} ({bd.get(0, 0):3d} / {bd.get(1, 0):3d} / {deep:3d})         | {nl:11d} | {d.split('(')[0]}")
