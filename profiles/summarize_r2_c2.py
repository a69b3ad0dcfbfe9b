#!/usr/bin/env python3
"""profiles/run_r2_c2.sh -> profiles/r2/c2_pmc_summary.json: counters of k_hmc_fused per launch (one launch =
100 transitions x 4096 chains) from summarize_r2_diag.py's summary.json, plus the derived VALU figures
bench.py --config c2 reports."""
import json, os, sys

src = sys.argv[1]
summ = json.load(open(os.path.join(src, "summary.json")))
name = [k for k in summ if "k_hmc_fused" in k][0]
k = summ[name]
waves, T, L = k["SQ_WAVES"], 100, 32
valu_pw = k["SQ_INSTS_VALU"] / waves / T
out = {
    "note": "rocprofv3, separate --pmc passes of `bench.py --config c2 --steps 5 --warmup 1` (profiles/run_r2_c2.sh); one launch = "
            "100 transitions x 4096 chains (one wavefront per chain, 4 per SIMD); FETCH_SIZE doubled (gfx950); SQ_* 'ACTIVE' / "
            "'CYCLES' counters tick in quad-cycles (one VALU instruction of a 64-lane wave = 4 cycles on a 16-lane SIMD)",
    "kernel": name, "per_launch": k,
    "derived": {
        "valu_instructions_per_wave_per_transition": valu_pw,
        "fp64_add_mul_fma_per_wave_per_transition":
            (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_FMA_F64"]) / waves / T,
        "salu_instructions_per_wave_per_transition": k["SQ_INSTS_SALU"] / waves / T,
        "leapfrog_fp64_instructions_per_transition": L * 2 * 6,
        "valu_busy_fraction_of_kernel_time_at_2.4GHz": k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4 * k["avg_ns"]),
        "valu_issue_ceiling_leapfrogs_per_s_at_this_instruction_count": 1024 * 2.4e9 / (4 * 4 * valu_pw) * 4 * L,
        "valu_issue_ceiling_leapfrogs_per_s_leapfrog_arithmetic_only": 1024 * 2.4e9 / (4 * 4 * L * 2 * 6) * 4 * L,
        "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch"),
    },
}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r2", "c2_pmc_summary.json")
json.dump(out, open(os.path.join(src, "c2_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
