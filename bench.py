#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec across all chains (BASELINE.json metric).

Default workload = BASELINE config c3 (the configuration the metric is quoted on):
1e4-dim correlated MVN, dense inverse mass matrix (fp64 MFMA path), NUTS
max_tree_depth=10, 4096 chains per GPU, synthetic inputs of SURVEY.md 8d.  A "step" is
one NUTS transition of every chain of the rank.  `--config c2` runs the 100-dim HMC
(L=32, diagonal mass) configuration instead.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N child ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Chains shard embarrassingly: each rank owns 4096 chains with its own seeds ("weak"
scaling); the only collective is the final gather of the last sample to rank 0 (RCCL over
xGMI), which is inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X FP64 matrix peak (AMD datasheet; = 256 CU * 4 SIMD * 32 flop/clk * 2.4 GHz).
#                               /opt/skills/guides/MI355X_MICROARCH.md lists no f64 MFMA row.
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


PEAK_FP64_VALU_TFLOPS = 78.6  # fp64 vector peak = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (datasheet)


_LIB_SHA = None


def lib_sha256():
    """sha256 of the libaehmc_hip.so this process loads: counter summaries are tied to the binary they measured."""
    global _LIB_SHA
    if _LIB_SHA is None:
        from aehmc_amd import _build, _lib
        _LIB_SHA = _build.library_hash(_lib.LIB_PATH)
    return _LIB_SHA


LINE_LIMIT = 7900  # (a margin under the ~8 080 bytes) the driver keeps about this much of stdout's tail: the WHOLE line, secondaries included, must fit


def compact(obj, verbose=False):
    """The JSON line without prose: `note` / `parity_note` strings (DESIGN.md section 4 has them) and null-valued keys of
    nested objects are dropped, floats keep 6 significant digits.  `--verbose` prints everything."""
    if verbose:
        return obj

    def walk(x, top):
        if isinstance(x, dict):
            out = {}
            for k, v in x.items():
                if k in ("note", "parity_note"):
                    continue
                if v is None and not top:
                    continue
                out[k] = walk(v, False)
            return out
        if isinstance(x, list):
            return [walk(v, False) for v in x]
        if isinstance(x, float):
            return float(f"{x:.6g}")
        return x
    return walk(obj, True)


def emit(obj, verbose=False):
    """Print THE line.  Should it still exceed LINE_LIMIT, the secondaries lose their least important keys first --
    never their value / frac -- so that the driver's record holds every number."""
    obj = compact(obj, verbose)
    line = json.dumps(obj, separators=(",", ":"))
    if not verbose and len(line) > LINE_LIMIT and obj.get("secondary"):
        keep = ("mfma_busy_fraction", "wait_any_fraction_of_wave_cycles", "wait_any_frac_of_wave_cycles",
                "valu_busy_fraction_of_kernel_time_at_2.4GHz", "l2_hit_rate", "clock_GHz_grbm")
        # `traffic_source` is always profiles/<round>/<config>_pmc_summary.json; long counter dicts shrink to the figures
        # the roofline argument uses; only then the prose (`workload`: the `config` names say which) and the rest
        for drop in ("<short counters_dropped>", "traffic_source", "<long counters>", "streaming_bytes_per_transition", "algorithmic_l2_bytes_per_launch",
                     "algorithmic_flops_per_launch", "whole_call_leapfrogs_per_s", "kernel", "issued_ops_per_elem", "workload", "counters", "hbm", "valu", "fp64_flops", "launches", "avg_launch_ms"):
            for e in obj["secondary"]:
                roof = e.get("roofline") if isinstance(e.get("roofline"), dict) else {}
                if drop == "<short counters_dropped>":  # (the top-level roofline keeps the full sentence)
                    for holder in (e, roof):
                        if holder.get("counters_dropped"):
                            holder["counters_dropped"] = "summary of another build"
                    continue
                if drop == "<long counters>":
                    for holder in (e, roof):
                        c = holder.get("counters")
                        if isinstance(c, dict) and len(c) > len(keep):
                            holder["counters"] = {k: v for k, v in c.items() if k in keep}
                    continue
                e.pop(drop, None)
                roof.pop(drop, None)
            line = json.dumps(obj, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    print(line)
    return line


PMC_STALE = {}  # summary name -> why its counters were NOT used in this run's line


PROFILES_DIR = os.environ.get("AEHMC_PROFILES_DIR") or os.path.join(ROOT, "profiles")  # where the summaries are looked up


def pmc_summary(name):
    """Counter summaries are collected OFFLINE (rocprofv3 --pmc cannot run inside the bench): the newest committed
    profiles/rN/<name>, and only if it was measured on the very binary this run loads -- every summary stores the
    sha256 of the libaehmc_hip.so it profiled (`lib_sha256`); one without it, or with another hash, is NOT used:
    peak / frac / traffic derived from it are dropped from the line and `counters_dropped` says why."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        path = os.path.join(PROFILES_DIR, rnd, name)
        if os.path.exists(path):
            summ = json.load(open(path))
            have = summ.get("lib_sha256")
            if have != lib_sha256():
                PMC_STALE[name] = (f"profiles/{rnd}/{name} was measured on another build of libaehmc_hip.so "
                                   f"(summary: {str(have)[:16]}, loaded: {lib_sha256()[:16]}): counter-derived "
                                   "peak / frac / traffic dropped")
                return None, None
            return summ, f"profiles/{rnd}/{name}"  # (separate rocprofv3 --pmc passes of the same workload, same lib_sha256)
    PMC_STALE[name] = f"no profiles/rN/{name}"
    return None, None


def build_c3(D, device, rho=0.5):
    """Sigma_ij = rho^|i-j| s_i s_j, s_i = 1 + (i mod 3); precision is tridiagonal
    analytically but stored and applied DENSE; imm = Sigma."""
    i = torch.arange(D, device=device)
    s = (1.0 + (i % 3)).to(torch.float64)
    R = rho ** (i[:, None] - i[None, :]).abs().to(torch.float64)
    Sigma = s[:, None] * R * s[None, :]
    del R
    d = torch.full((D,), (1 + rho * rho) / (1 - rho * rho), dtype=torch.float64, device=device)
    d[0] = d[-1] = 1.0 / (1 - rho * rho)
    P = torch.diag(d)
    off = torch.full((D - 1,), -rho / (1 - rho * rho), dtype=torch.float64, device=device)
    P += torch.diag(off, 1) + torch.diag(off, -1)
    P = P / s[:, None] / s[None, :]
    Sigma = 0.5 * (Sigma + Sigma.T)
    P = 0.5 * (P + P.T)
    return Sigma.contiguous(), P.contiguous()


def other_configs():
    """The other single-GPU BASELINE configs (c1, c2, c5) as further `secondary` entries of the default
    line, so that their numbers are in the driver's record too: each runs in a child process of its own
    (`bench.py --config cN --no-cpu-baseline`, default steps), after this process's own measurements."""
    out = []
    for cfg, extra in (("c2", []), ("c2", ["--fp-contract"]), ("c5", []), ("c1", [])):
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK",
                                                                     "TORCHELASTIC_RUN_ID")}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--no-cpu-baseline"] + extra,
                               capture_output=True, text=True, timeout=300, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            out.append({"config": cfg + ("-fp_contract" if extra else ""), "workload": d["config"]["workload"],
                        "value": d["value"], "unit": d["unit"],
                        "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                        "roofline": d.get("roofline")})
        except Exception as e:  # a failing side measurement must not cost the main line
            out.append({"config": cfg, "error": repr(e)[:300]})
    return out


HMC_PER_CALL = 20


def diag_case(kind, D, C, device):
    """Secondary workloads (diagonal / scalar mass: HBM-bound): D-dim isotropic Gaussian,
    diagonal inverse mass matrix of ones, eps = 0.5 D^-1/4, C chains; `kind` nuts (default depth) or
    hmc (L = 32).  Shared with tools/diag_run.py, whose rocprofv3 --pmc passes count the bytes."""
    from aehmc_amd import RandomStream, hmc, nuts, targets
    eps = 0.5 * D ** -0.25
    q0 = torch.as_tensor(np.random.default_rng(0).standard_normal((C, D)), device=device)
    imm = torch.ones(D, dtype=torch.float64, device=device)
    tgt = targets.IsoGaussian()
    mod = nuts if kind == "nuts" else hmc
    kernel = mod.new_kernel(RandomStream(seeds=list(range(C))), tgt)
    state = mod.new_state(q0, tgt)
    if kind == "nuts":
        return state, (lambda st: kernel(st, eps, imm))
    # HMC: HMC_PER_CALL transitions per engine call (kernel.sample, the user-level scan): the position stays on chip
    # for the transitions of a call; n_leapfrog is the call's total
    return state, (lambda st: (kernel.sample(st, eps, imm, 32, HMC_PER_CALL, keep_samples=False)[1], None))


def bench_secondary(eng, device, steps, warmup, D=10_000, C=4096):
    """Diagonal-mass NUTS and HMC at the headline shape.  `achieved` = HBM bytes per transition as
    COUNTED by rocprofv3 (FETCH_SIZE x 2 + WRITE_SIZE of the kernels of one transition, separate
    --pmc passes of the same workload: the newest profiles/rN/diag_pmc_summary.json) / the transition's kernel
    time measured here with HIP events on the launch stream."""
    pmc, pmc_src = pmc_summary("diag_pmc_summary.json") if (D, C) == (10_000, 4096) else (None, None)
    out = []
    for kind, main_kernel, fc in (("nuts", "k_nuts_wide", 0), ("hmc", "k_hmc_wide", 0), ("hmc", "k_hmc_wide", 1)):
        eng.set_option("fp_contract", fc)
        pkey = kind + ("_fp_contract" if fc else "")  # section of the counter summary
        state, step = diag_case(kind, D, C, device)
        for _ in range(warmup):
            info, _ = step(state)
            state = info.state._replace(momentum=None)
        eng.profile_enable(True)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        nl = torch.zeros((), dtype=torch.int64, device=device)
        for _ in range(steps):
            info, _ = step(state)
            state = info.state._replace(momentum=None)
            nl += info.n_leapfrog.sum()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        kern_ms, kern_n, _ = eng.profile_read()
        eng.profile_enable(False)
        nl = int(nl.item())
        per_call = HMC_PER_CALL if kind == "hmc" else 1  # transitions per engine call (= per HIP-event pair)
        traffic = None
        if pmc and pkey in pmc:
            traffic = pmc[pkey]["hbm_bytes_per_transition"]
        avg_ms = kern_ms / max(kern_n, 1) / per_call
        alg = (48.0 if kind == "hmc" else 88.0) * D * nl / (steps * per_call)  # SURVEY 8d streaming figure, for reference only
        hbm = {"unit": "GB/s", "peak": PEAK_HBM_GBS, "traffic": traffic,
               "achieved": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
               "streaming_bytes_per_transition": alg,
               "note": "counted HBM bytes per transition / kernel time; the chain state is on chip, so the counted bytes are far "
                       "below the streaming figure (48 D / 88 D per leapfrog)"}
        hbm["frac"] = hbm["achieved"] / PEAK_HBM_GBS if traffic else None
        if kind == "nuts":  # bound by HBM on its counted bytes (checkpoints, proposal copies, parked trajectory ends)
            roof = dict(hbm, bound="hbm", kernel=main_kernel + " (+ k_draw_momentum)", avg_launch_ms=avg_ms,
                        launches=kern_n * per_call, traffic_source=pmc_src if traffic else None,
                        counters_dropped=PMC_STALE.get("diag_pmc_summary.json"))
        else:
            # the position stays in registers for the transitions of a call: HBM sees the normals in and little else, and
            # the bound is fp64 VALU issue -- 6 UNFUSED operations per element and leapfrog (the reference rounds every
            # product and sum), against 16 lanes x 4 SIMDs x 256 CUs x 2.4 GHz lane-operations per second
            # (fp_contract = 1: 2 fused operations per element and leapfrog -- the same trajectory in a third of the
            #  issue slots; `achieved` then counts the operations that mode executes)
            # ONE operation count for both modes (VERDICT r5 hygiene 15): the 6 D algorithmic operations of the leapfrog, so
            # that the faster mode shows the higher fraction; what fp_contract = 1 actually issues is `issued_ops_per_elem`
            peak_ops = 256 * 4 * 16 * 2.4e9 / 1e12
            per_elem = 2.0 if fc else 6.0
            ops = nl / dt * 6.0 * D / 1e12
            roof = {"bound": "valu", "unit": "Tlane-op/s (fp64, 6 D per leapfrog)", "achieved": ops,
                    "peak": peak_ops, "frac": ops / peak_ops, "issued_ops_per_elem": per_elem,
                    "kernel": main_kernel + " (+ k_draw_momentum)", "avg_launch_ms": avg_ms, "launches": kern_n * per_call,
                    "traffic": traffic, "traffic_source": pmc_src if traffic else None, "hbm": hbm,
                    "counters_dropped": PMC_STALE.get("diag_pmc_summary.json"),
                    "note": f"6 D fp64 operations per leapfrog in the integration ({per_elem:.0f} D issued in this mode); the momentum draw "
                            "(PCG64 + ziggurat, one wavefront per chain) is VALU-bound as well and is not counted in `achieved`"}
        out.append({"config": f"diag-{kind}" + ("-fp_contract" if fc else ""),
                    "workload": f"{D}-dim isotropic Gaussian, diagonal mass, {'NUTS depth 10' if kind == 'nuts' else 'HMC L=32'}, "
                                f"{C} chains" + (", engine option fp_contract=1 (1e-6 relative, not bit parity)" if fc else ""),
                    "value": nl / dt, "unit": "leapfrog-steps/s", "ms_per_transition": dt / (steps * per_call) * 1e3,
                    "leapfrogs_per_transition": nl / (steps * per_call), "roofline": roof})
    eng.set_option("fp_contract", 0)
    return out


def dense_mid_secondary(eng, device, C=4096, T=10):
    """Mid-size dense problems (shared dense inverse mass matrix AND dense-precision target, D = 100 and 200: the
    block-resident kernels of csrc/nuts_block_reg.cuh / nuts_block_roll.cuh, one launch per sample(T) call).  Per leapfrog
    and chain two D x D products on fp64 MFMA inside the workgroup = 4 D^2 flop (algorithmic, metrics.py:71 and the
    target's P r); `achieved` = that / the call's time, against the fp64 MFMA peak; the MFMA-busy counter of the same
    workload (profiles/r4/dense/mid200_pmc_summary.json) is quoted when it was measured on the loaded library."""
    from aehmc_amd import RandomStream, nuts, targets
    out = []
    for D in (100, 200):
        try:
            r = np.random.default_rng(0)

            def spd():
                A = r.normal(size=(D, D))
                M = A @ A.T / D + np.eye(D)
                return 0.5 * (M + M.T)
            P, imm = spd(), torch.as_tensor(spd(), device=device)
            tgt = targets.DenseMVN(torch.zeros(D, dtype=torch.float64, device=device), torch.as_tensor(P, device=device))
            q0 = torch.as_tensor(r.standard_normal((C, D)), device=device)
            kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            eps = 0.3 * D ** -0.25
            info = kernel.sample(nuts.new_state(q0, tgt), eps, imm, 3)[1]
            best, nl = None, 0
            for _ in range(3):  # (the first call at a new T also allocates its [T, C, D] sample buffer)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                info = kernel.sample(info.state._replace(momentum=None), eps, imm, T)[1]
                torch.cuda.synchronize(device)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, nl = dt, int(info.n_leapfrog.sum().item())
            tflops = nl * 4.0 * D * D / best / 1e12
            roof = {"bound": "mfma", "unit": "TFLOP/s", "achieved": tflops, "peak": PEAK_FP64_MFMA_TFLOPS,
                    "frac": tflops / PEAK_FP64_MFMA_TFLOPS, "kernel": "k_nuts_block_reg" if D < 192 else "k_nuts_block_roll",
                    "launches": 1, "avg_launch_ms": best * 1e3, "traffic": None,
                    "note": "4 D^2 flop per leapfrog and chain; every chain of a 16-chain workgroup steps through its own "
                            "tree, so the products of a round also carry the rows of chains that have finished"}
            name = os.path.join("dense", f"mid{D}_pmc_summary.json")
            pmc, src = pmc_summary(name)
            if pmc:  # (separate rocprofv3 --pmc passes of tools/debug/mid_dense.py D 4096 10: the same call)
                dv = pmc["derived"]
                roof["traffic"] = dv.get("hbm_bytes_per_launch")
                roof["traffic_source"] = src
                roof["counters"] = {"mfma_busy_fraction": dv["mfma_busy_fraction"],
                                    "wait_any_fraction_of_wave_cycles": dv["wait_any_frac"]}
            roof["counters_dropped"] = PMC_STALE.get(name)
            out.append({"config": f"dense-nuts-d{D}",
                        "workload": f"{D}-dim correlated MVN (dense precision), dense inverse mass matrix, NUTS depth 10, {C} chains, "
                                    f"sample({T}) in one launch",
                        "value": nl / best, "unit": "leapfrog-steps/s", "ms_per_transition": best / T * 1e3,
                        "leapfrogs_per_transition": nl / T, "roofline": roof})
        except Exception as e:  # a failing side measurement must not cost the main line
            out.append({"config": f"dense-nuts-d{D}", "error": repr(e)[:300]})
    return out


def pc_dense_secondary(eng, device, D=200, C=4096, T=10):
    """One dense inverse mass matrix PER CHAIN (what window_adaptation.run(is_mass_matrix_full=True) returns), D = 200,
    coordinate-wise target: csrc/nuts_pc_dense.cuh, one launch per sample(T) call, the wavefront that owns a chain streams
    its matrix once per leapfrog (linear dense mode) + three times at the start of a transition.  Bound: HBM on those bytes
    (D^2 x 8 per product and chain); `traffic` = the counted bytes of the same call when a stamped summary exists."""
    from aehmc_amd import PerChain, RandomStream, nuts, targets
    try:
        g = torch.Generator(device=device).manual_seed(0)
        A = torch.randn(C, D, D, dtype=torch.float64, device=device, generator=g)
        imm = torch.baddbmm(0.3 * torch.eye(D, dtype=torch.float64, device=device).expand(C, D, D), A, A.transpose(1, 2), alpha=1.0 / D)
        imm = 0.5 * (imm + imm.transpose(1, 2))
        del A
        r = np.random.default_rng(0)
        tgt = targets.DiagGaussian(r.normal(size=D), 0.5 + r.random(D))
        q0 = torch.as_tensor(r.standard_normal((C, D)), device=device)
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
        eps, pi = 0.25 * D ** -0.25, PerChain(imm)
        info = kernel.sample(nuts.new_state(q0, tgt), eps, pi, 2)[1]
        best, nl = None, 0
        for _ in range(3):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            info = kernel.sample(info.state._replace(momentum=None), eps, pi, T)[1]
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, nl = dt, int(info.n_leapfrog.sum().item())
        alg = (nl + 3.0 * C * T) * D * D * 8.0
        roof = {"bound": "hbm", "unit": "GB/s", "achieved": alg / best / 1e9, "peak": PEAK_HBM_GBS,
                "frac": alg / best / 1e9 / PEAK_HBM_GBS, "kernel": "k_nuts_pc_dense", "launches": 1, "avg_launch_ms": best * 1e3,
                "algorithmic_bytes_per_launch": alg, "traffic": None,
                "note": "D^2 x 8 bytes per product and chain: one product per leapfrog + three per transition"}
        name = os.path.join("dense", "pc200_pmc_summary.json")
        pmc, src = pmc_summary(name)
        if pmc:
            # the profiled run's launches are sample(2) and sample(10) calls: its counted RATE x this launch's duration
            rate = pmc["derived"].get("hbm_GBs")
            roof["traffic"] = rate * 1e9 * best if rate else pmc["derived"].get("hbm_bytes_per_launch")
            roof["traffic_source"] = src
        roof["counters_dropped"] = PMC_STALE.get(name)
        return [{"config": f"pc-dense-nuts-d{D}",
                 "workload": f"{D}-dim Gaussian, one dense inverse mass matrix per chain, NUTS depth 10, {C} chains, sample({T}) in one launch",
                 "value": nl / best, "unit": "leapfrog-steps/s", "ms_per_transition": best / T * 1e3,
                 "leapfrogs_per_transition": nl / T, "roofline": roof}]
    except Exception as e:  # a failing side measurement must not cost the main line
        return [{"config": f"pc-dense-nuts-d{D}", "error": repr(e)[:300]}]


def custom_secondary(eng, device, D=5000, C=4096, T=4):
    """A user-defined target given by its log-DENSITY only (Student-t; differentiated by the engine, csrc/dual.cuh) on the
    workgroup-per-chain NUTS kernel compiled at run time, beside the built-in diagonal Gaussian at the same shape
    (VERDICT r4 item 7).  The pass of that kernel is bound by VALU issue, so the ratio is the density's arithmetic."""
    from aehmc_amd import RandomStream, nuts, targets
    src = """
template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
  const double hn = prm[0][i], inv_s = prm[1][i], inv_nu = prm[2][i];
  const T z = q * inv_s;
  return hn * log1p(z * z * inv_nu);
}
"""
    try:
        r = np.random.default_rng(0)
        nu, sg = 3.0 + 5 * r.random(D), 0.5 + r.random(D)
        q0 = torch.as_tensor(r.standard_normal((C, D)), device=device)
        imm = torch.ones(D, dtype=torch.float64, device=device)
        eps, out = 0.4 * D ** -0.25, {}
        for name, tgt in (("custom", targets.Custom(src, params=[-0.5 * (nu + 1.0), 1.0 / sg, 1.0 / nu])),
                          ("builtin", targets.DiagGaussian(np.zeros(D), sg))):
            kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt)
            state = nuts.new_state(q0, tgt)
            for _ in range(2):
                state = kernel(state, eps, imm)[0].state._replace(momentum=None)
            torch.cuda.synchronize(device)
            t0, nl = time.perf_counter(), 0
            for _ in range(T):
                info = kernel(state, eps, imm)[0]
                state = info.state._replace(momentum=None)
                nl += int(info.n_leapfrog.sum().item())
            torch.cuda.synchronize(device)
            out[name] = nl / (time.perf_counter() - t0)
        return [{"config": f"custom-student-t-nuts-d{D}",
                 "workload": f"{D}-dim Student-t defined by its log-density (engine-differentiated), diagonal mass, NUTS depth 10, {C} chains",
                 "value": out["custom"], "unit": "leapfrog-steps/s", "builtin_diag_gaussian": out["builtin"],
                 "builtin_over_custom": out["builtin"] / out["custom"], "kernel": "k_nuts_wide<512,16,LDS,custom> (hipRTC)",
                 # the pass is bound by VALU issue: 6 operations of the leapfrog + 9 of the density and its derivative (one of
                 # them the log1p, ~25 issued instructions) per element and leapfrog, against the fp64 lane-operation rate
                 "roofline": {"bound": "valu", "unit": "Tlane-op/s (fp64, 15 D per leapfrog)",
                              "achieved": out["custom"] * 15.0 * D / 1e12, "peak": 256 * 4 * 16 * 2.4e9 / 1e12,
                              "frac": out["custom"] * 15.0 * D / 1e12 / (256 * 4 * 16 * 2.4e9 / 1e12), "launches": T}}]
    except Exception as e:  # a failing side measurement must not cost the main line
        return [{"config": f"custom-student-t-nuts-d{D}", "error": repr(e)[:300]}]


def traced_secondary(eng, device, D=1000, C=4096, T=5):
    """Neal's funnel written as a PYTHON logprob_fn (round 6: traced by aehmc_amd/tracing.py, differentiated in ONE reverse
    sweep whose loops run over the lanes of the chain's wavefront) on the one-launch joint kernel -- VERDICT r5 item 5's
    shape (D = 1000, 4096 chains; the hand-written HIP template in forward mode: 1.0e6 leapfrog/s).  The kernel runs the
    lock-step engine's own stage / bookkeeping functions on L2 / HBM rows: 88 D bytes per leapfrog and chain (SURVEY 8d's
    streaming NUTS figure) against the HBM peak."""
    from aehmc_amd import RandomStream, nuts, targets
    try:
        def funnel(q):
            v, x = q[0], q[1:]
            return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()

        r = np.random.default_rng(D)
        tgt = targets.from_callable(funnel, D)
        q0 = torch.as_tensor(0.3 * r.standard_normal((C, D)), device=device)
        imm = torch.ones(D, dtype=torch.float64, device=device)
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        state = kernel.sample(state, 0.05, imm, 2, keep_samples=False)[1].state._replace(momentum=None)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        _, info, _, _ = kernel.sample(state, 0.05, imm, T, keep_samples=False)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum().item())
        gbs = nl / dt * 88.0 * D / 1e9
        return [{"config": f"python-funnel-nuts-d{D}",
                 "workload": f"{D}-dim Neal's funnel as a Python function (traced, reverse-mode gradient), NUTS depth 6, {C} chains",
                 "value": nl / dt, "unit": "leapfrog-steps/s", "ms_per_transition": dt / T * 1e3,
                 "kernel": "k_nuts_joint_rows (hipRTC) + generated aehmc_logp_grad",
                 "roofline": {"bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS,
                              "traffic": None, "launches": 1}}]
    except Exception as e:  # a failing side measurement must not cost the main line
        return [{"config": f"python-funnel-nuts-d{D}", "error": repr(e)[:300]}]


def glm_secondary(eng, device, N=100_000, D=8, C=1024, T=4):
    """Logistic regression over N data rows given by its per-row log-likelihood only (targets.CustomGLM; VERDICT r5 item 4's
    shape: N = 1e5, D = 8, 1024 chains, NUTS depth 6; round 5: 8.0e4 leapfrog/s): a workgroup of eight wavefronts per chain
    sweeps the rows (k_nuts_glm_wg).  Bound by VALU issue on the row function; counted at 4 D + 60 fp64 lane-operations per
    row (dot product and gradient accumulation; one exponential, the log(1 + e) series and a division for softplus and its
    derivative)."""
    from aehmc_amd import RandomStream, nuts, targets
    src = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
    name = f"custom-logistic-nuts-n{N}"
    try:
        r = np.random.default_rng(0)
        X = r.normal(size=(N, D))
        w = r.normal(size=D)
        y = (r.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
        tgt = targets.CustomGLM(src, torch.as_tensor(X, device=device), torch.as_tensor(y, device=device))
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        state = nuts.new_state(torch.as_tensor(w + 0.1 * r.standard_normal((C, D)), device=device), tgt)
        eps, imm = 0.3 / np.sqrt(N), torch.ones(D, dtype=torch.float64, device=device)
        for _ in range(2):
            state = kernel(state, eps, imm)[0].state._replace(momentum=None)
        torch.cuda.synchronize(device)
        t0, nl = time.perf_counter(), 0
        for _ in range(T):
            info = kernel(state, eps, imm)[0]
            state = info.state._replace(momentum=None)
            nl += int(info.n_leapfrog.sum().item())
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        ops = nl / dt * N * (4.0 * D + 60.0) / 1e12
        peak = 256 * 4 * 16 * 2.4e9 / 1e12
        return [{"config": name,
                 "workload": f"logistic regression by its row log-likelihood, N={N} rows, D={D}, NUTS depth 6, {C} chains",
                 "value": nl / dt, "unit": "leapfrog-steps/s", "ms_per_transition": dt / T * 1e3,
                 "kernel": f"k_nuts_glm_wg<{D},8> (hipRTC)",
                 "roofline": {"bound": "valu", "unit": "Tlane-op/s (fp64, 4 D + 60 per row)", "achieved": ops, "peak": peak,
                              "frac": ops / peak, "launches": T}}]
    except Exception as e:  # a failing side measurement must not cost the main line
        return [{"config": name, "error": repr(e)[:300]}]


def launch_ranks(n, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child ranks (one process per
    GPU, RCCL rendezvous on 127.0.0.1) and relay rank 0's JSON line.  The parent never touches
    the GPU -- children are new processes, nothing is exec'ed over an initialised HIP runtime.
    The children are polled: the first rank that exits non-zero (or a wall-clock timeout,
    AEHMC_BENCH_TIMEOUT seconds, default 3600) ends the run -- its siblings, which would otherwise
    wait for it in a collective for ever, are terminated, the failing rank's stderr tail is relayed
    and the exit status is non-zero."""
    import tempfile
    if timeout_s is None:
        timeout_s = float(os.environ.get("AEHMC_BENCH_TIMEOUT", "3600"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        err = tempfile.TemporaryFile(mode="w+")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out, stderr=err, text=True))

    def tail(f, nbytes=3000):
        f.flush()
        f.seek(0)
        return f.read()[-nbytes:]

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with code {rcs[bad[0]]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = f"timeout after {timeout_s:.0f} s (ranks still running: {[r for r, rc in enumerate(rcs) if rc is None]})"
            bad = [r for r, rc in enumerate(rcs) if rc is None][:1]
            break
        time.sleep(0.2)
    if failed:
        stop_all()
        for r in bad:
            sys.stderr.write(f"bench.py: ---- stderr tail of rank {r} ----\n{tail(logs[r][1])}\n")
        sys.exit(f"bench.py: {failed}; sibling ranks terminated")
    sys.stdout.write(tail(logs[0][0], 1 << 22))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 5 (c1, c2: 20; c5: 1000 transitions -- as many draws as "
                                                            "warm-up steps -- in one launch)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 1 (c1, c2: 5 -- their steps are < 1 ms each, and the "
                                                             "clocks of an idle GPU take longer than that to come up)")
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs; c4 = c3 with 32768 chains split over the ranks (strong scaling)")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: 4096; c5: 1024 = 8192 over 8 GPUs; "
                                                             "c4: 32768 / ranks)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the diagonal-mass secondary lines")
    ap.add_argument("--verbose", action="store_true", help="print the line with its explanatory `note` strings (longer than "
                                                           "the driver's stdout tail keeps)")
    ap.add_argument("--fp-contract", action="store_true",
                    help="engine option fp_contract=1: fast arithmetic in the leapfrog bodies of the register-resident HMC "
                         "kernels (1e-6 relative instead of bit parity with the oracle; c2 and the diagonal-mass HMC line)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c1": 20, "c2": 20, "c5": 1000}.get(args.config, 5)
    if args.warmup is None:
        args.warmup = {"c1": 5, "c2": 5, "c5": 1000}.get(args.config, 1)
    if args.chains is None:
        args.chains = {"c5": 1024}.get(args.config, 4096)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()  # counts devices without initialising the GPU
        if have < args.gpus and os.environ.get("AEHMC_BENCH_ONE_DEVICE") != "1":
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
        return launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AEHMC_BENCH_FAIL_RANK") == str(rank) and world > 1:  # tests: a rank that dies early
        sys.exit("bench.py: injected failure of this rank (AEHMC_BENCH_FAIL_RANK)")
    # dry-run knobs for a 1-GPU box: AEHMC_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 and
    # AEHMC_DIST_BACKEND=gloo swaps RCCL for gloo (the driver's multi-GPU runs use neither)
    if os.environ.get("AEHMC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    dist_backend = None
    # (a single rank started by torch.distributed.run still joins its one-rank group: the same RCCL
    #  code path as the multi-GPU runs, which a 1-GPU box can exercise)
    if world > 1 or os.environ.get("TORCHELASTIC_RUN_ID"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = dist_backend = os.environ.get("AEHMC_DIST_BACKEND", "nccl")
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("AEHMC_BENCH_PG_TIMEOUT", "900")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")

    from aehmc_amd import RandomStream, hmc, nuts, targets
    from aehmc_amd.engine import get_engine
    from aehmc_amd.parallel import barrier, gather_samples, max_over_ranks, shard_checksums, sum_over_ranks

    strong = args.config == "c4"
    if strong:  # config 4: 32768 chains sharded across the GPUs of the node
        args.config, args.chains = "c3", 32768 // world
    if args.config == "c1":
        return bench_c1(args)
    if args.config == "c5":
        return bench_c5(args, rank, world, device)
    C = args.chains
    eng = get_engine(device)
    eng.set_option("fp_contract", 1 if args.fp_contract else 0)
    if os.environ.get("AEHMC_BENCH_ONE_DEVICE") == "1" and world > 1:
        eng.set_option("streamk", 0)  # ranks share the device: the persistent stream-K grid is not co-resident
    seeds = [1000 + rank * C + c for c in range(C)]
    q0 = np.random.default_rng(1234 + rank).standard_normal((C, args.dim or (100 if args.config == "c2" else 10_000)))
    D = q0.shape[1]

    if args.config == "c3":
        Sigma, P = build_c3(D, device)
        mu = torch.zeros(D, dtype=torch.float64, device=device)
        target = targets.DenseMVN(mu, P)
        imm = Sigma
        eps = 0.5 * D ** -0.25
        kernel = nuts.new_kernel(RandomStream(seeds=seeds), target, max_num_expansions=10)
        state = nuts.new_state(torch.as_tensor(q0, device=device), target)
        step = lambda st: kernel(st, eps, imm)
        workload = (f"c3: {D}-dim correlated MVN (AR(1) rho=0.5, dense precision), dense inverse mass "
                    f"matrix, NUTS max_tree_depth=10, {C} chains/GPU, eps={eps:.4f}")
    else:
        target = targets.IsoGaussian()
        imm = torch.ones(D, dtype=torch.float64, device=device)
        eps, L = 0.1, 32
        kernel = hmc.new_kernel(RandomStream(seeds=seeds), target)
        state = hmc.new_state(torch.as_tensor(q0, device=device), target)
        NT = 100  # transitions per engine call (100 transitions; one launch on the fused path)
        step = lambda st: (kernel.sample(st, eps, imm, L, NT, keep_samples=False)[1], None)
        workload = (f"c2: {D}-dim isotropic Gaussian, HMC L={L}, diagonal mass, {C} chains/GPU, eps={eps}; "
                    f"one step = {NT} transitions of every chain"
                    + ("; engine option fp_contract=1 (fused multiply-adds and merged half kicks in the leapfrog loop: "
                       "1e-6 relative, not bit parity)" if args.fp_contract else ""))

    # warm-up runs exactly what a timed step runs (including the leapfrog tally and one gather), so
    # that no lazily loaded code object lands inside the timed region
    n_leap = torch.zeros((), dtype=torch.int64, device=device)
    for _ in range(args.warmup):
        info, _ = step(state)
        state = info.state._replace(momentum=None)
        n_leap += info.n_leapfrog.sum()
    if args.warmup:
        gather_samples(state.position)

    eng.profile_enable(True)
    n_leap.zero_()
    barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info, _ = step(state)
        state = info.state._replace(momentum=None)
        n_leap += info.n_leapfrog.sum()
    torch.cuda.synchronize(device)
    t_g = time.perf_counter()
    gathered = gather_samples(state.position)  # the path's one exchange step: to rank 0
    torch.cuda.synchronize(device)
    t_g = time.perf_counter() - t_g
    barrier(device)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, device)
    total_leap = sum_over_ranks(int(n_leap.item()), device)
    ranks_seen = sum_over_ranks(1, device)
    kern_ms, kern_n, kern_flops = eng.profile_read()
    eng.profile_enable(False)
    assert rank != 0 or gathered.shape[0] == C * world
    # the gather moved every rank's rows bit for bit: per-rank checksums of the bit patterns against the same sums
    # over the shards of the gathered array (outside the timed region)
    sums = shard_checksums(state.position)
    rows_match = None
    if rank == 0:
        got = torch.stack([gathered[r * C:(r + 1) * C].contiguous().view(torch.int64).sum() for r in range(world)])
        rows_match = bool(torch.equal(got.cpu(), sums.cpu()))

    if rank != 0:
        return
    value = total_leap / elapsed
    if args.config == "c3":
        # algorithmic flops per launch = 2 * live_rows * D * D (live rows counted on the device;
        # SURVEY.md 8d's per-leapfrog figure is one such row per mat-vec), / mean launch duration
        flops = kern_flops / max(kern_n, 1)
        avg_s = kern_ms / 1e3 / max(kern_n, 1)
        achieved = flops / avg_s / 1e12
        traffic, traffic_src = None, None
        pj, src = pmc_summary("c3_pmc_summary.json")
        if pj and D == 10_000 and C == 4096:
            traffic = pj["gemm_summary"]["traffic_bytes_per_launch_avg"]
            traffic_src = src + "; FETCH_SIZE x 2 + WRITE_SIZE"
        roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                    "counters_dropped": PMC_STALE.get("c3_pmc_summary.json"),
                    "algorithmic_bytes_per_launch": (2.0 * flops / (2.0 * D * D) * D + D * D) * 8,
                    "kernel": "gemm_nt_f64_streamk_kernel", "avg_launch_ms": avg_s * 1e3, "launches": kern_n,
                    "avg_rows_per_launch": flops / (2.0 * D * D), "gemm_share_of_step_time": kern_ms / 1e3 / elapsed,
                    "parity_note": "c3's arithmetic is the dense branch, which no reference value pins (SURVEY.md 8c): parity "
                                   "is HIP == C restatement, checked at this depth in tests/test_gpu_configs.py, and c3 == the isotropic "
                                   "D=1e4 problem on the diagonal kernels under q' = chol(Sigma) q (same file)"}
    else:
        # fused HMC kernel (100 transitions per launch, state in registers): HBM sees only the I/O of a launch
        # (idle); the bound is fp64 VALU issue.  achieved / peak are leapfrog rates: peak = the issue ceiling at the
        # kernel's COUNTED instructions per transition (rocprofv3 SQ_INSTS_VALU), 1024 SIMDs x 2.4 GHz /
        # (4 waves x 4 cycles x instructions per wave and transition).
        avg_s = kern_ms / 1e3 / max(kern_n, 1)
        c2_name = "c2_fc_pmc_summary.json" if args.fp_contract else "c2_pmc_summary.json"
        pj, src = pmc_summary(c2_name)
        if not (pj and D == 100 and C == 4096):
            pj, src = None, None
        flop_rate = value / world * 9.0 * D / 1e12  # ~9 D flop per leapfrog
        # `frac` is against SURVEY.md 8d's FIXED ceiling -- 9 D flop per leapfrog against the fp64 vector peak -- so that it
        # measures efficiency; the ceiling derived from the kernel's own counted instructions (how stall-free the issue is)
        # is reported beside it as valu.issue_ceiling / valu.issue_frac (round 4 printed that one as `frac`).
        roofline = {"bound": "valu", "unit": "TFLOP/s", "kernel": "k_hmc_fused", "avg_launch_ms": avg_s * 1e3,
                    "launches": kern_n, "achieved": flop_rate, "peak": PEAK_FP64_VALU_TFLOPS,
                    "frac": flop_rate / PEAK_FP64_VALU_TFLOPS, "flop_per_leapfrog": 9.0 * D, "traffic": None,
                    "traffic_source": src, "counters_dropped": PMC_STALE.get(c2_name),
                    "note": "9 D flop per leapfrog (SURVEY.md 8d) against the fp64 vector peak"}
        if pj:
            d = pj["derived"]
            ceil = d["valu_issue_ceiling_leapfrogs_per_s_at_this_instruction_count"]
            roofline.update({
                "traffic": d["hbm_bytes_per_launch"],
                "hbm": {"achieved": d["hbm_bytes_per_launch"] / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": d["hbm_bytes_per_launch"] / avg_s / 1e9 / PEAK_HBM_GBS,
                        "note": "q, dU/dq in and out once per launch of 100 transitions: HBM is idle"},
                "valu": {"issue_ceiling": ceil, "issue_frac": value / world / ceil, "unit": "leapfrog-steps/s",
                         "instructions_per_wave_per_transition": d["valu_instructions_per_wave_per_transition"],
                         "leapfrog_fp64_instructions_per_transition": d["leapfrog_fp64_instructions_per_transition"],
                         "busy_fraction_rocprof": d["valu_busy_fraction_of_kernel_time_at_2.4GHz"],
                         "note": "one wavefront per chain, 4 per SIMD; a 64-lane fp64 VALU instruction occupies its 16-lane "
                                 "SIMD for 4 cycles; issue_ceiling = 1024 SIMDs x 2.4 GHz / (4 waves x 4 cycles x counted "
                                 "instructions per wave and transition)"}})

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args.config, D, q0, target, imm, eps)
    secondary = None
    if args.config == "c3" and world == 1 and not args.no_secondary and D == 10_000:
        del state, info, kernel, target, imm, gathered
        torch.cuda.empty_cache()
        secondary = bench_secondary(eng, device, max(args.steps, 3), 2)
        secondary += dense_mid_secondary(eng, device)
        secondary += pc_dense_secondary(eng, device)
        secondary += custom_secondary(eng, device)
        secondary += traced_secondary(eng, device)
        secondary += glm_secondary(eng, device)
        torch.cuda.empty_cache()
        secondary += other_configs()

    emit({
        "metric": "leapfrog-steps/sec across all chains", "value": value, "unit": "leapfrog-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "chains_total": C * world, "dim": D, "fp_contract": int(args.fp_contract),
                   "lib_sha256": lib_sha256(),
                   "leapfrogs_per_step": total_leap / args.steps, "ranks_seen": ranks_seen,
                   "gather": {"to": "rank 0", "bytes": (world - 1) * C * D * 8, "ms": t_g * 1e3,
                              "backend": dist_backend, "rows_match_ranks_bitwise": rows_match}},
        "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary}, args.verbose)


def bench_c1(args):
    """Config c1 (plumbing check): README example, one chain, NUTS, eps=1e-2 -- the reference
    runtime (Aesara C backend) is unavailable; the value must equal README.md:53-54."""
    from aehmc_amd import RandomStream, nuts, targets
    from aehmc_amd.engine import get_engine
    eng = get_engine()
    target = targets.StdNormal()
    times, pos, nl = [], None, 0
    for i in range(args.warmup + args.steps):
        if i == args.warmup:
            eng.profile_enable(True)
        kernel = nuts.new_kernel(RandomStream(seed=0), target)
        state = nuts.new_state(0.0, target)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info, _ = kernel(state, 1e-2, 1.0)
        pos = info.state.position.item()
        times.append(time.perf_counter() - t0)
        nl = int(info.n_leapfrog.item())
    kern_ms, kern_n, _ = eng.profile_read()
    eng.profile_enable(False)
    dt = sum(times[args.warmup:]) / max(args.steps, 1)
    # One wavefront walks the 136 leapfrogs of the tree one dependent instruction after the other: neither HBM nor the
    # matrix cores bound it, the ISSUE LATENCY of a lone wavefront does.  peak = clock / (issue cycles per leapfrog at
    # the COUNTED instruction mix: 4 cycles per 64-lane VALU instruction on its 16-lane SIMD, one per scalar / LDS /
    # memory / branch instruction; rocprofv3 counters of this very launch, profiles/r4/c1_pmc_summary.json);
    # achieved = leapfrogs / the kernel's launch duration (HIP events on the launch stream).
    avg_s = kern_ms / 1e3 / max(kern_n, 1)
    pj, src = pmc_summary("c1_pmc_summary.json")
    roofline = {"bound": "latency", "unit": "leapfrog-steps/s", "kernel": "k_nuts_resident<64, 1> (one wavefront)",
                "avg_launch_ms": avg_s * 1e3, "launches": kern_n, "achieved": nl / avg_s if avg_s > 0 else None,
                "peak": None, "frac": None, "traffic": None, "traffic_source": src,
                "counters_dropped": PMC_STALE.get("c1_pmc_summary.json"),
                "whole_call_leapfrogs_per_s": nl / dt,
                "note": "latency roofline of a single wavefront: issue cycles per leapfrog at the counted instruction mix; the "
                        "whole call adds ~0.06 ms of host work and the read-back to the kernel"}
    if pj and roofline["achieved"]:
        d = pj["derived"]
        roofline.update({"peak": d["latency_roofline_leapfrogs_per_s_at_2.4GHz"],
                         "frac": roofline["achieved"] / d["latency_roofline_leapfrogs_per_s_at_2.4GHz"],
                         "issue_cycles_per_leapfrog": d["issue_cycles_per_leapfrog_lower_bound"],
                         "valu_instructions_per_leapfrog": d["valu_instructions_per_leapfrog"],
                         "salu_instructions_per_leapfrog": d["salu_instructions_per_leapfrog"],
                         "clock_GHz_during_the_profiled_launch": d.get("clock_GHz_grbm")})
    emit({
        "metric": "leapfrog-steps/sec across all chains", "value": nl / dt, "unit": "leapfrog-steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "c1: README example, 1-D standard normal, NUTS, step_size=1e-2, single chain",
                   "position": pos, "matches_readme_value": pos == 1.1034719409361107, "leapfrogs": nl,
                   "lib_sha256": lib_sha256()},
        "roofline": roofline, "cpu_baseline": None}, args.verbose)


def bench_c5(args, rank, world, device):
    """Config c5: regression of examples/LinearRegression.ipynb scaled to 1e5 rows, D=2, NUTS +
    window adaptation, 8192 chains over 8 GPUs (1024 per rank by default here)."""
    from aehmc_amd import RandomStream, nuts, targets, window_adaptation
    from aehmc_amd.parallel import barrier, gather_samples, max_over_ranks, sum_over_ranks
    C = args.chains
    rng = np.random.default_rng(0)
    N = 100_000
    X = rng.normal(0, 1, size=(N,))
    y = 3 * X + rng.normal(0, 1)
    target = targets.LinearRegression(X, y)
    q0 = np.array([3.0, np.log(0.5)]) + 0.05 * np.random.default_rng(1 + rank).normal(size=(C, 2))
    kernel = nuts.new_kernel(RandomStream(seeds=[5000 + rank * C + c for c in range(C)]), target)
    state = nuts.new_state(torch.as_tensor(q0, device=device), target)
    from aehmc_amd.engine import get_engine
    eng = get_engine(device)
    t_w = time.perf_counter()
    state, (eps, imm), _ = window_adaptation.run(kernel, state, max(args.warmup, 20))
    torch.cuda.synchronize(device)
    t_w = time.perf_counter() - t_w
    eng.profile_enable(True)
    barrier(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    _, info, _, _ = kernel.sample(state, eps, imm, args.steps, keep_samples=False)
    gathered = gather_samples(info.state.position)
    torch.cuda.synchronize(device)
    barrier(device)
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    total_rank = int(info.n_leapfrog.sum().item())
    total = sum_over_ranks(total_rank, device)
    chains_total = sum_over_ranks(C, device)
    kern_ms, kern_n, _ = eng.profile_read()
    eng.profile_enable(False)
    if rank == 0:
        assert gathered.shape[0] == chains_total
        rate = total / elapsed
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_c5(X, y, info, eps, imm)
        # The dominant kernel is k_nuts_linreg (the whole sample() call is ONE launch of it).  Neither HBM nor
        # MFMA bounds it: per leapfrog a chain needs 3 fp64 FMAs (6 flop) per data row, the rows (16 B each) come
        # from L2 / LDS once per leapfrog round of a 4-chain workgroup.  achieved = algorithmic flop of the launch
        # (6 x rows x this rank's leapfrogs) / its duration (HIP events on the launch stream); counters offline.
        avg_s = kern_ms / 1e3 / max(kern_n, 1)
        flops = 6.0 * N * total_rank / max(kern_n, 1)
        achieved = flops / avg_s / 1e12
        pj, src = pmc_summary("c5_pmc_summary.json")
        if not (pj and C == 1024 and pj.get("run", {}).get("transitions") == args.steps):  # counters of THIS launch shape only
            pj, src = None, None
        roofline = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP64_VALU_TFLOPS, "kernel": "k_nuts_linreg", "avg_launch_ms": avg_s * 1e3,
                    "launches": kern_n, "algorithmic_flops_per_launch": flops,
                    "algorithmic_l2_bytes_per_launch": 16.0 * N * total_rank / 4 / max(kern_n, 1),
                    "traffic": pj["derived"]["hbm_bytes_per_launch"] if pj else None, "traffic_source": src,
                    "counters": pj["derived"] if pj else None, "counters_dropped": PMC_STALE.get("c5_pmc_summary.json"),
                    "note": "3 FMAs per row, chain and leapfrog against the fp64 vector peak; the sweep of a 4-chain "
                            "workgroup also pulls 16 B per row through its CU's 64 B/clk vector-memory path (rows beyond "
                            "the 10176 kept in LDS), which bounds a sweep at about the same time as the FMAs: DESIGN.md"}
        emit({
            "metric": "leapfrog-steps/sec across all chains", "value": rate, "unit": "leapfrog-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"c5: linear regression, {N} rows, D=2, NUTS after {max(args.warmup, 20)} "
                                   f"window-adaptation steps ({t_w:.2f} s, one launch), {C} chains/GPU",
                       "chains_total": chains_total, "data_rows_per_s": rate * N,
                       "leapfrogs_per_step": total / args.steps},
            "roofline": roofline, "cpu_baseline": cpu}, args.verbose)


def cpu_baseline_c5(X, y, info, eps, imm):
    """c5 on the host: the C restatement runs NUTS transitions of the first chains from the GPU's
    post-warm-up state (one thread, then OpenMP over chains, one chain per thread).  The oracle takes
    one step size / metric per call, so the chains share the MEDIAN of the adapted per-chain values."""
    from oracle import c_oracle as co
    cores = os.cpu_count() or 1
    otgt = co.Target(co.T_LINREG, 2, X=X, y=y)
    pos = info.state.position.cpu().numpy()
    e = float(np.median(eps.value.cpu().numpy()))
    metric = co.Metric(np.median(imm.value.cpu().numpy().reshape(-1, 2), axis=0), 2)

    def run(n, threads, reps):
        rng = co.site_states([9000 + c for c in range(n)], 4)
        q, U, g = co.new_state(otgt, pos[:n].copy())
        t0, nl = time.perf_counter(), 0
        for _ in range(reps):
            nl += int(co.nuts_step(otgt, metric, rng, e, q, U, g, nthreads=threads)["n_leapfrog"].sum())
        return nl, time.perf_counter() - t0

    # bounded samples of a few seconds each: a leapfrog is one pass over the 1e5 rows (~1e6 flop), so 4 chains x 3000
    # transitions (~5e4 leapfrogs) on one thread, and one chain per thread x 3000 transitions on at most 64 threads
    # (more threads than that only measure the fork / join of the OpenMP team on this workload)
    reps = 3000
    n1, dt1 = run(4, 1, reps)
    used = min(cores, 64, pos.shape[0])
    na, dta = run(used, used, reps)
    return {"value": na / dta, "unit": "leapfrog-steps/s", "cores": used, "kind": "port",
            "sample": f"{used} chains x {reps} NUTS transitions (median adapted step size / metric), OpenMP over "
                      f"chains ({na} leapfrogs, {dta:.1f} s)",
            "single_thread": {"value": n1 / dt1, "unit": "leapfrog-steps/s", "cores": 1,
                              "sample": f"4 chains x {reps} NUTS transitions ({n1} leapfrogs, {dt1:.1f} s)"},
            "host_cpu_count": cores,
            "note": "C restatement of aehmc semantics (oracle/c), not Aesara; reported, not optimised"}


def cpu_baseline(config, D, q0, target, imm, eps):
    """The C restatement (oracle, "port") timed on this host's cores on a bounded sample, twice:
    one thread, and all cores (OpenMP over chains, one chain per thread) -- SURVEY.md 8d."""
    from oracle import c_oracle as co
    cores = os.cpu_count() or 1

    def run(n, threads, max_exp=None):
        seeds = [1000 + c for c in range(n)]
        if config == "c3":
            rng = co.site_states(seeds, 4)
            q, U, g = co.new_state(otgt, q0[:n].copy())
            t0 = time.perf_counter()
            res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp, nthreads=threads)
            return int(res["n_leapfrog"].sum()), time.perf_counter() - t0
        rng = co.site_states(seeds, 2)
        q, U, g = co.new_state(otgt, q0[:n].copy())
        t0 = time.perf_counter()
        for _ in range(reps):
            co.hmc_step(otgt, metric, rng, eps, 32, q, U, g, nthreads=threads)
        return n * 32 * reps, time.perf_counter() - t0

    if config == "c3":
        otgt = co.Target(co.T_DENSE_MVN, D, mu=np.zeros(D), prec=target.precision.cpu().numpy())
        metric = co.Metric(imm.cpu().numpy(), D)
        # bounded sample: ONE chain through a whole depth-10 transition on one thread (~35-70 leapfrogs), and an
        # all-cores leg truncated at max_num_expansions=4 (2 + 3 + 5 + 9 = 19 leapfrogs per chain; a full tree is
        # ~57).  Every mat-vec streams an 800 MB matrix, so the all-cores leg is DRAM-bound well below the core
        # count: 32 chains (= threads that get work) saturate it within the time budget.
        n_all = min(cores, 32, q0.shape[0])
        what = ("NUTS transition truncated at max_num_expansions=4 (19 leapfrogs/chain), same D / target / dense "
                "metric as the GPU run")
        nl1, dt1 = run(1, 1, 10)
        used = min(cores, n_all)
        nla, dta = run(n_all, used, 4)
        return {"value": nla / dta, "unit": "leapfrog-steps/s", "cores": used, "cores_of_host": f"{used} of {cores}", "kind": "port",
                "sample": f"{n_all} chains x {what}, {used} OpenMP threads over chains ({nla} leapfrogs, {dta:.1f} s)",
                "single_thread": {"value": nl1 / dt1, "unit": "leapfrog-steps/s", "cores": 1,
                                  "sample": f"1 chain x one full NUTS transition at max_tree_depth=10 ({nl1} leapfrogs, "
                                            f"{dt1:.1f} s)"},
                "host_cpu_count": cores,
                "note": "C restatement of aehmc semantics (oracle/c), not Aesara; reported, not optimised"}
    else:
        otgt, metric = co.Target(co.T_ISO_GAUSSIAN, D), co.Metric(np.ones(D), D)
        # (one oracle call = one transition of all chains: ~1e8 flop, so the fork / join of a 256-thread
        #  team would dominate -- all chains of the config, at most 32 threads)
        reps, n_all = 2000, q0.shape[0]  # ~1e10 flop-ish per call: a few seconds on 32 threads
        cores = min(cores, 32)
        what = f"{reps} HMC transitions (L=32) per chain"
    n_one = 8
    reps_all, reps = reps, max(reps // 10, 1)  # the single-thread leg: 8 chains, a tenth of the transitions
    nl1, dt1 = run(n_one, 1)
    reps = reps_all
    used = min(cores, n_all)  # threads that get a chain
    nla, dta = run(n_all, used)
    return {"value": nla / dta, "unit": "leapfrog-steps/s", "cores": used, "kind": "port",
            "sample": f"{n_all} chains x {what}, {used} OpenMP threads over chains ({nla} leapfrogs, {dta:.1f} s)",
            "single_thread": {"value": nl1 / dt1, "unit": "leapfrog-steps/s", "cores": 1,
                              "sample": f"8 chains x {reps_all // 10} HMC transitions (L=32) per chain ({nl1} leapfrogs, {dt1:.1f} s)"},
            "host_cpu_count": cores,
            "note": "C restatement of aehmc semantics (oracle/c), not Aesara; reported, not optimised"}


if __name__ == "__main__":
    main()
