#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of profiles/run_r1_profiles.sh (FETCH_SIZE, WRITE_SIZE;
counter_collection CSVs under gpurun_out/pmc_fetch and gpurun_out/pmc_write) into
profiles/<round>/c3_pmc_summary.json.

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KB; on
gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The counters sit on the L2's
memory side: Infinity-Cache hits are included.
usage: python profiles/summarize_pmc.py [gpurun_out] [profiles/r1]"""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r1"
D = 10_000


def per_kernel(path, counter):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "avg_KB_per_launch": sum(v) / len(v), "max_KB": max(v)} for k, v in agg.items()}


out = {"FETCH_SIZE": per_kernel(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"),
       "WRITE_SIZE": per_kernel(os.path.join(src, "pmc_write"), "WRITE_SIZE")}
name = next(k for k in out["FETCH_SIZE"] if "gemm_nt_f64_streamk_kernel" in k)
f, w = out["FETCH_SIZE"][name], out["WRITE_SIZE"][name]
out["gemm_summary"] = {
    "kernel": name, "launches": f["launches"],
    "fetch_bytes_avg_raw": f["avg_KB_per_launch"] * 1024,
    "fetch_bytes_avg_x2_gfx950": 2 * f["avg_KB_per_launch"] * 1024,
    "write_bytes_avg": w["avg_KB_per_launch"] * 1024,
    "traffic_bytes_per_launch_avg": (2 * f["avg_KB_per_launch"] + w["avg_KB_per_launch"]) * 1024,
    "traffic_bytes_full_size_launch": (2 * f["max_KB"] + w["max_KB"]) * 1024,
    "algorithmic_bytes_full_size_launch": (2 * 4096 * D + D * D) * 8,
    "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0`; averages are over "
            "launches with the live row count (compaction); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; "
            "counts L2-miss requests incl. Infinity-Cache hits"}
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "c3_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out["gemm_summary"], indent=1))

# c2: the fused HMC kernel (100 transitions of 4096 chains per launch)
if os.path.isdir(os.path.join(src, "pmc_fetch_c2")):
    f2, w2 = per_kernel(os.path.join(src, "pmc_fetch_c2"), "FETCH_SIZE"), per_kernel(os.path.join(src, "pmc_write_c2"), "WRITE_SIZE")
    name = next(k for k in f2 if "k_hmc_fused" in k)
    c2 = {"FETCH_SIZE": f2, "WRITE_SIZE": w2,
          "hmc_fused_summary": {
              "kernel": name, "launches": f2[name]["launches"],
              "fetch_bytes_avg_x2_gfx950": 2 * f2[name]["avg_KB_per_launch"] * 1024,
              "write_bytes_avg": w2[name]["avg_KB_per_launch"] * 1024,
              "traffic_bytes_per_launch_avg": (2 * f2[name]["avg_KB_per_launch"] + w2[name]["avg_KB_per_launch"]) * 1024,
              "io_bytes_per_launch_expected": 4096 * 100 * 8 * (2 + 2) + 100 * 4096 * 12,
              "note": "one launch = 100 transitions x 4096 chains, state in registers: expected I/O is q, g in and out once "
                      "(13 MB) plus the acceptance / divergence history (4.9 MB); FETCH_SIZE doubled as above (the kernel's "
                      "8-byte-per-lane loads are outside the guide's calibration)"}}
    json.dump(c2, open(os.path.join(dst, "c2_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(c2["hmc_fused_summary"], indent=1))
