#!/usr/bin/env python3
"""profiles/run_r6.sh -> gpurun_out/r6_summaries/*_pmc_summary.json (copied to profiles/r6/ when committed).

Per workload directory gpurun_out/r6_<tag>: kernel-trace stats (launch count, average duration) and the PMC counters
averaged per launch, per kernel; then the derived figures bench.py divides by its own timings, in the layouts it reads.
EVERY summary carries `lib_sha256`, the sha256 of the libaehmc_hip.so that was profiled (written by run_r6.sh next to
the counters): bench.py uses a summary only when that equals the library it has loaded.

Counter units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950
tallies 128-byte requests at 64 B); SQ_*CYCLES / SQ_WAIT* / SQ_ACTIVE* in quad-cycles; GRBM_GUI_ACTIVE summed over 8 XCDs.
usage: summarize_r6.py <gpurun_out> <tag> [<tag> ...]"""
import collections
import csv
import glob
import json
import os
import sys

root, tags = sys.argv[1], sys.argv[2:]
dst = os.path.join(root, "r6_summaries")
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fold(src, last_dispatch_of=None):
    """per kernel: calls, avg_ns (stats) + counters averaged per launch (or, `last_dispatch_of`: of the LAST dispatch of
    the kernels whose name contains that string)"""
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"].split("(")[0]
            out[k].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]),
                          pct=float(r["Percentage"]))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if last_dispatch_of:
            sel = [r for r in rows if last_dispatch_of in r["Kernel_Name"]]
            if sel:
                last = max(int(r["Dispatch_Id"]) for r in sel)
                rows = [r for r in rows if last_dispatch_of not in r["Kernel_Name"] or int(r["Dispatch_Id"]) == last]
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:  # a counter may come in several rows per dispatch (per XCD / shader engine): sum them
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, cn), per in agg.items():
            out[k][cn] = sum(per.values()) / len(per)
            out[k].setdefault("pmc_launches", len(per))
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
            if "avg_ns" in d:
                d["hbm_GBs"] = d["hbm_bytes_per_launch"] / d["avg_ns"]
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if "SQ_INSTS_SALU" in d and "SQ_WAVES" in d:
            d["salu_per_wave"] = d["SQ_INSTS_SALU"] / d["SQ_WAVES"]
        if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
            d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in d and "avg_ns" in d:
            d["clock_GHz_grbm"] = d["GRBM_GUI_ACTIVE"] / 8 / d["avg_ns"]
        if "SQ_ACTIVE_INST_VALU" in d and "avg_ns" in d:
            d["valu_busy_fraction_at_2.4GHz"] = d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4 * d["avg_ns"])
    return out


def find(out, sub):
    ks = sorted((k for k in out if sub in k), key=lambda k: -out[k].get("total_ns", 0))
    return (ks[0], out[ks[0]]) if ks else (None, None)


def meta(src):
    sha = open(os.path.join(src, "lib_sha256.txt")).read().strip()
    cmd = open(os.path.join(src, "command.txt")).read().strip()
    return sha, cmd


def save(name, doc):
    json.dump(doc, open(os.path.join(dst, name), "w"), indent=1)
    print("wrote", name)


NOTE = ("rocprofv3, kernel-trace stats + SEPARATE --pmc passes of `{cmd}` (profiles/run_r6.sh); averages per launch; "
        "FETCH_SIZE doubled (gfx950); SQ cycle counters in quad-cycles")

diag = {}
for tag in tags:
    src = os.path.join(root, "r6_" + tag)
    if not os.path.isdir(src):
        continue
    sha, cmd = meta(src)
    if tag in ("c2", "c2_fc"):
        out = fold(src)
        name, k = find(out, "k_hmc_fused")
        waves, T, L = k["SQ_WAVES"], 100, 32
        valu_pw = k["SQ_INSTS_VALU"] / waves / T
        per_leap = 2 * (2 if tag == "c2_fc" else 6)  # fp64 operations per leapfrog of the wave's 2 elements per lane
        save(tag + "_pmc_summary.json", {
            "note": NOTE.format(cmd=cmd) + "; one launch = 100 transitions x 4096 chains (one wavefront per chain, 4 per SIMD)",
            "lib_sha256": sha, "kernel": name, "per_launch": k,
            "derived": {
                "valu_instructions_per_wave_per_transition": valu_pw,
                "fp64_add_mul_fma_per_wave_per_transition":
                    (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_FMA_F64"]) / waves / T,
                "salu_instructions_per_wave_per_transition": k["SQ_INSTS_SALU"] / waves / T,
                "leapfrog_fp64_instructions_per_transition": L * per_leap,
                "valu_busy_fraction_of_kernel_time_at_2.4GHz": k["valu_busy_fraction_at_2.4GHz"],
                "wait_any_fraction_of_wave_cycles": k.get("wait_any_frac"),
                "clock_GHz_grbm": k.get("clock_GHz_grbm"),
                # 1024 SIMDs x 2.4 GHz / (4 waves x 4 cycles x instructions per wave and transition) x 4 chains x L
                "valu_issue_ceiling_leapfrogs_per_s_at_this_instruction_count": 1024 * 2.4e9 / (4 * 4 * valu_pw) * 4 * L,
                "valu_issue_ceiling_leapfrogs_per_s_leapfrog_arithmetic_only": 1024 * 2.4e9 / (4 * 4 * L * per_leap) * 4 * L,
                "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch"),
            }})
    elif tag.startswith("diag_"):
        from bench import HMC_PER_CALL
        out = fold(src)
        kind = "nuts" if tag == "diag_nuts" else "hmc"
        key = "k_nuts_wide" if kind == "nuts" else "k_hmc_wide"
        per = 1 if kind == "nuts" else HMC_PER_CALL
        (_, main), (_, dm) = find(out, key), find(out, "k_draw_momentum")
        diag.setdefault("lib_sha256", sha)
        assert diag["lib_sha256"] == sha, "the diag workloads were profiled with different libraries"
        diag[kind + ("_fp_contract" if tag.endswith("_fc") else "")] = {
            "command": cmd, "hbm_bytes_per_transition": (main["hbm_bytes_per_launch"] + dm["hbm_bytes_per_launch"]) / per,
            "transitions_per_launch": per, "kernel_ms_per_transition": (main["avg_ns"] + dm["avg_ns"]) / per / 1e6,
            "main_kernel": main, "k_draw_momentum": dm}
    elif tag == "c3":
        out = fold(src)
        name, g = find(out, "gemm_nt_f64_streamk_kernel<true, 8")
        summ = {"kernel": name, "launches": g.get("pmc_launches"), "traffic_bytes_per_launch_avg": g["hbm_bytes_per_launch"],
                "fetch_bytes_avg_x2_gfx950": 2 * g["FETCH_SIZE"] * 1024, "write_bytes_avg": g["WRITE_SIZE"] * 1024,
                "avg_launch_ms_rocprof": g["avg_ns"] / 1e6, "clock_GHz_grbm": g.get("clock_GHz_grbm")}
        if "TCC_REQ_sum" in g:
            summ.update(l2_requests_per_launch=g["TCC_REQ_sum"], l2_hits=g["TCC_HIT_sum"], l2_misses=g["TCC_MISS_sum"],
                        l2_hit_rate=g["TCC_HIT_sum"] / max(g["TCC_HIT_sum"] + g["TCC_MISS_sum"], 1),
                        l2_request_bytes_per_launch=g["TCC_REQ_sum"] * 128)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
            summ["mfma_busy_fraction"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g["GRBM_GUI_ACTIVE"] / 8)
        keep = ("calls", "avg_ns", "pct", "hbm_bytes_per_launch", "hbm_GBs", "clock_GHz_grbm")
        save("c3_pmc_summary.json", {
            "note": NOTE.format(cmd=cmd) + "; averages over the GEMM launches of one c3 transition (live rows shrink along it)",
            "lib_sha256": sha, "gemm_summary": summ,
            "other_kernels": {k: {a: b for a, b in d.items() if a in keep}
                              for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("total_ns", 0))[:10] if k != name}})
    elif tag == "c5":
        out = fold(src, last_dispatch_of="k_nuts_linreg")
        run = None
        for line in open(os.path.join(src, "stats.log")):
            if line.startswith("{") and "leapfrogs" in line:
                run = json.loads(line)
        name, k = find(out, "k_nuts_linreg")
        # the sample() launch is the LAST dispatch of the kernel: its duration from the kernel trace
        t = None
        for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True):
            rows = sorted((r for r in csv.DictReader(open(f)) if "k_nuts_linreg" in r["Kernel_Name"]),
                          key=lambda r: int(r["Start_Timestamp"]))
            t = (int(rows[-1]["End_Timestamp"]) - int(rows[-1]["Start_Timestamp"])) * 1e-9
        nl, N = run["leapfrogs"], run["rows"]
        d = {"leapfrogs": nl, "launch_ms": t * 1e3, "leapfrogs_per_s_kernel": nl / t,
             "fp64_fma_tflops_algorithmic": 6.0 * N * nl / t / 1e12, "frac_of_fp64_vector_peak_78.6": 6.0 * N * nl / t / 78.6e12}
        if "hbm_bytes_per_launch" in k:
            d["hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
        if "SQ_INSTS_VALU_FMA_F64" in k:
            d["fma_f64_per_row_chain_leapfrog"] = k["SQ_INSTS_VALU_FMA_F64"] * 64 / (N * nl)
            d["fma_f64_tflops_counted"] = k["SQ_INSTS_VALU_FMA_F64"] * 64 * 2 / t / 1e12
            d["fma_share_of_valu"] = k["SQ_INSTS_VALU_FMA_F64"] / k["SQ_INSTS_VALU"]
        d["valu_instructions_per_wave"] = k.get("valu_per_wave")
        d["wait_any_frac_of_wave_cycles"] = k.get("wait_any_frac")
        if "SQ_ACTIVE_INST_VALU" in k:
            d["valu_busy_fraction_of_kernel_time_at_2.4GHz"] = k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e9 * t)
        if "TCC_REQ_sum" in k:
            d["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1)
            d["l2_TBs_at_128B_per_request"] = k["TCC_REQ_sum"] * 128 / t / 1e12
        if "GRBM_GUI_ACTIVE" in k:
            d["clock_GHz_grbm"] = k["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
        save("c5_pmc_summary.json", {"note": NOTE.format(cmd=cmd) + "; values of the LAST k_nuts_linreg dispatch = the sample() "
                                     "launch after the warm-up launch", "lib_sha256": sha, "run": run, "per_launch": k, "derived": d})
    elif tag == "c1":
        out = fold(src)
        name, k = find(out, "k_nuts_resident")
        NL = 136  # leapfrogs of the README transition (G1)
        valu, salu = k["SQ_INSTS_VALU"] / NL, k["SQ_INSTS_SALU"] / NL
        other = sum(k.get(c, 0.0) for c in ("SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                            "SQ_INSTS_BRANCH")) / NL
        # a lone wavefront issues its instructions one after the other: a 64-lane VALU instruction occupies its 16-lane
        # SIMD for 4 cycles (fp64 transcendentals / divisions are sequences of such instructions and are counted as
        # such), a scalar / LDS / memory / branch instruction takes at least one issue cycle
        cycles = 4 * valu + salu + other
        clock = k.get("clock_GHz_grbm") or 2.4
        save("c1_pmc_summary.json", {
            "note": NOTE.format(cmd=cmd) + "; ONE wavefront runs the 136-leapfrog transition of the README example",
            "lib_sha256": sha, "kernel": name, "per_launch": k,
            "derived": {
                "leapfrogs_per_launch": NL, "kernel_us_per_launch": k["avg_ns"] / 1e3,
                "valu_instructions_per_leapfrog": valu, "salu_instructions_per_leapfrog": salu,
                "other_instructions_per_leapfrog": other,
                "fp64_add_mul_fma_trans_per_leapfrog": sum(k.get(c, 0.0) for c in (
                    "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) / NL,
                "issue_cycles_per_leapfrog_lower_bound": cycles,
                "clock_GHz_grbm": k.get("clock_GHz_grbm"),
                "latency_roofline_leapfrogs_per_s_at_2.4GHz": 2.4e9 / cycles,
                "latency_roofline_leapfrogs_per_s_at_measured_clock": clock * 1e9 / cycles,
                "achieved_leapfrogs_per_s_kernel": NL / (k["avg_ns"] * 1e-9),
                "frac_of_latency_roofline_at_2.4GHz": NL / (k["avg_ns"] * 1e-9) / (2.4e9 / cycles),
                "wait_any_fraction_of_wave_cycles": k.get("wait_any_frac")}})
    else:  # generic (mid200: the block-resident dense kernel)
        out = fold(src)
        keep = sorted(out.items(), key=lambda kv: -kv[1].get("total_ns", 0))[:6]
        doc = {"note": NOTE.format(cmd=cmd), "lib_sha256": sha, "kernels": dict(keep)}
        name, k = find(out, "k_nuts_pc_dense" if tag.startswith("pc") else "k_nuts_block")
        if k and "GRBM_GUI_ACTIVE" in k:
            doc["derived"] = {"kernel": name, "clock_GHz_grbm": k.get("clock_GHz_grbm"), "valu_per_wave": k.get("valu_per_wave"),
                              "wait_any_frac": k.get("wait_any_frac"), "avg_launch_ms": k["avg_ns"] / 1e6,
                              "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch"), "hbm_GBs": k.get("hbm_GBs")}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
                doc["derived"]["mfma_busy_fraction"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * k["GRBM_GUI_ACTIVE"] / 8)
            if "TCC_HIT_sum" in k:
                doc["derived"]["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1)
        save(tag + "_pmc_summary.json", doc)

if len(diag) > 1:
    from bench import HMC_PER_CALL
    diag["note"] = ("rocprofv3, kernel-trace stats + separate --pmc passes of tools/diag_run.py (profiles/run_r6.sh) = bench.py's "
                    "secondary workloads (D = 1e4 diagonal mass, 4096 chains); FETCH_SIZE doubled (gfx950); averages per launch; "
                    f"NUTS transition = k_draw_momentum + k_nuts_wide; HMC: one k_draw_momentum + one k_hmc_wide launch per engine "
                    f"call of {HMC_PER_CALL} transitions")
    # a partial re-run keeps the other sections of an existing summary measured on the SAME library
    prev = os.path.join(dst, "diag_pmc_summary.json")
    if os.path.exists(prev):
        old = json.load(open(prev))
        if old.get("lib_sha256") == diag["lib_sha256"]:
            for k, v in old.items():
                diag.setdefault(k, v)
    save("diag_pmc_summary.json", diag)
