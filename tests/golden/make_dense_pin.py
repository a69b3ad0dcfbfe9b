#!/usr/bin/env python3
"""A SECOND, independent derivation of one dense-metric NUTS transition (and one HMC transition), D = 3.

Why: the headline configuration (c3) lives entirely in the dense branch of gaussian_metric
(/root/reference/aehmc/metrics.py:52-59), for which the reference publishes no RNG-dependent value -- the
oracle (oracle/np_oracle.py, oracle/c/aehmc_oracle.c) is pinned there only through the exact unit tables
of tests/test_metrics.py and an invariance argument.  This script does NOT import oracle/ or aehmc_amd: it
is written from SURVEY.md Appendix A (the language-neutral restatement) as one flat procedure with the
dense literals spelled out --

    L    = numpy.linalg.cholesky(imm)                                   # metrics.py:56
    S    = scipy.linalg.solve_triangular(L, eye(D), lower=True, trans=1)  # metrics.py:58  (= L^-T)
    p    = S @ z,  z = Generator.normal(0, 1, size=D)                    # metrics.py:66-67
    K(p) = 0.5 * dot(imm @ p, p)                                         # metrics.py:70-73
    turn = dot(imm @ p_l, rho) <= 0  |  dot(imm @ p_r, rho) <= 0         # metrics.py:94-102

-- and the RNG of scheme A (one numpy Generator per call site, children of SeedSequence(seed) in creation
order).  Its outputs are committed as tests/golden/dense_pin_v1.json; tests/test_dense_pin.py checks both
restatements (CPU) and the HIP path (GPU) against them.  Run: python tests/golden/make_dense_pin.py
"""
import json
import os

import numpy as np
from scipy.linalg import solve_triangular

HERE = os.path.dirname(os.path.abspath(__file__))


def logaddexp(a, b):
    return float(np.logaddexp(a, b))


def nuts_transition(seed, q, mu, P, imm, eps, max_exp, thr=1000.0):
    D = len(q)
    g1, g2, g3, g4 = [np.random.default_rng(s) for s in np.random.SeedSequence(seed).spawn(4)]
    L = np.linalg.cholesky(imm)
    S = solve_triangular(L, np.eye(D), lower=True, trans=1)

    def pot(x):  # U = 0.5 r^T P r, grad = P r (P symmetric)
        r = x - mu
        gr = P @ r
        return 0.5 * float(r @ gr), gr

    def K(p):
        return 0.5 * float((imm @ p) @ p)

    def turning(pl, pr, psum):
        rho = psum - (pr + pl) / 2
        return bool((float((imm @ pl) @ rho) <= 0) | (float((imm @ pr) @ rho) <= 0))

    def leap(x, p, gr, h):  # integrators.py:54-73
        p = p - (0.5 * h) * gr
        x = x + (1 * h) * (imm @ p)
        U, gr = pot(x)
        p = p - (0.5 * h) * gr
        return x, p, U, gr

    def bern(gen, pr):
        return bool(gen.binomial(1, pr))

    U0, gr0 = pot(q)
    p0 = S @ g1.normal(0, 1, size=D)
    H0 = U0 + K(p0)
    prop = dict(x=q, p=p0, U=U0, g=gr0, E=H0, w=0.0, slpa=-np.inf)
    left = right = (q, p0, U0, gr0)
    psum = p0.copy()
    ckp, cks = np.zeros((max_exp, D)), np.zeros((max_exp, D))
    tmin = tmax = 0
    n_leap, trace = 0, []
    out = None
    for j in range(max_exp):
        go_right = bern(g2, 0.5)
        d = 1.0 if go_right else -1.0
        x, p, U, gr = right if go_right else left

        def gen_prop(x, p, U, gr):
            E = U + K(p)
            delta = H0 - E
            if np.isnan(delta):
                delta = -np.inf
            return dict(x=x, p=p, U=U, g=gr, E=E, w=delta, slpa=min(delta, 0.0)), bool(abs(delta) > thr)

        # first step of the sub-trajectory: outside the loop, no turning check, stale checkpoint indices
        x, p, U, gr = leap(x, p, gr, d * eps)
        n_leap += 1
        sub, div0 = gen_prop(x, p, U, gr)
        ssum = p.copy()
        ckp[tmax], cks[tmax] = p, ssum  # step 0 is even
        first = (dict(sub), (x, p, U, gr), ssum.copy(), 1, div0, False)
        length, div, term = 1, div0, False
        cur = (x, p, U, gr)
        for step in range(1, 2 ** j + 1):
            x, p, U, gr = leap(cur[0], cur[1], cur[3], d * eps)
            if not div0:
                n_leap += 1
            new, div = gen_prop(x, p, U, gr)
            pa = 1.0 / (1.0 + np.exp(-(new["w"] - sub["w"])))
            if np.isnan(pa):
                pa = 0.0
            take = bern(g3, pa)
            merged = dict(new if take else sub)
            merged["w"] = logaddexp(sub["w"], new["w"])
            merged["slpa"] = logaddexp(sub["slpa"], new["slpa"])
            sub = merged
            ssum = ssum + p
            n1 = 0
            while (step >> n1) & 1:
                n1 += 1
            tmax = bin(step >> 1).count("1")
            tmin = tmax - n1 + 1
            if step % 2 == 0:
                ckp[tmax], cks[tmax] = p, ssum
            term = False
            if tmax >= tmin:
                for i in range(tmax, tmin - 1, -1):
                    if turning(ckp[i], p, ssum - cks[i] + ckp[i]):
                        term = True
                        break
            cur = (x, p, U, gr)
            length += 1
            if div or term:
                break
        if div0:  # trajectory.py:336: the first-step tuple is returned (the scan above still drew from g3)
            sub, cur, ssum, length, div, term = first
        if go_right:
            right = cur
        else:
            left = cur
        psum = psum + ssum
        acc_prob = float(np.exp(sub["slpa"])) / length
        pb = min(max(float(np.exp(sub["w"] - prop["w"])), 0.0), 1.0)
        take_b = bern(g4, pb)  # drawn either way
        if div or term:
            prop = dict(prop, slpa=logaddexp(sub["slpa"], prop["slpa"]))
        else:
            chosen = dict(sub if take_b else prop)
            chosen["w"] = logaddexp(prop["w"], sub["w"])
            chosen["slpa"] = logaddexp(prop["slpa"], sub["slpa"])
            prop = chosen
        is_turn = turning(left[1], right[1], psum)
        trace.append(dict(direction=int(go_right), length=int(length), proposal=[float(v) for v in prop["x"]]))
        out = dict(position=prop["x"], momentum=prop["p"], U=prop["U"], grad=prop["g"], acceptance_probability=acc_prob,
                   num_doublings=j + 1, is_diverging=bool(div), is_turning=bool(is_turn))
        if div or is_turn or term:
            break
    out["n_leapfrog"] = n_leap
    out["initial_momentum"] = p0
    out["trace"] = trace
    return out


def hmc_transition(seed, q, mu, P, imm, eps, nsteps, thr=1000.0):
    D = len(q)
    g1, g2 = [np.random.default_rng(s) for s in np.random.SeedSequence(seed).spawn(2)]
    L = np.linalg.cholesky(imm)
    S = solve_triangular(L, np.eye(D), lower=True, trans=1)
    r = q - mu
    gr = P @ r
    U0 = 0.5 * float(r @ gr)
    p0 = S @ g1.normal(0, 1, size=D)
    x, p, U = q, p0, U0
    for _ in range(nsteps):
        p = p - (0.5 * eps) * gr
        x = x + (1 * eps) * (imm @ p)
        r = x - mu
        gr = P @ r
        U = 0.5 * float(r @ gr)
        p = p - (0.5 * eps) * gr
    p = -1.0 * p
    delta = (U0 + 0.5 * float((imm @ p0) @ p0)) - (U + 0.5 * float((imm @ p) @ p))
    if np.isnan(delta):
        delta = -np.inf
    pa = min(max(float(np.exp(delta)), 0.0), 1.0)
    acc = bool(g2.binomial(1, pa))
    return dict(position=x if acc else q, U=U if acc else U0, acceptance_probability=pa, accepted=acc,
                is_diverging=bool(abs(delta) > thr), initial_momentum=p0)


def tolist(d):
    return {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def main():
    r = np.random.default_rng(2024)
    cases = []
    for name, seed, eps, max_exp in (("nuts-dense-a", 7, 0.08, 10), ("nuts-dense-b", 8, 0.13, 10), ("nuts-dense-c", 12, 0.05, 10), ("nuts-dense-cut", 9, 0.15, 3)):
        A, B = r.normal(size=(3, 3)), r.normal(size=(3, 3))
        P = A @ A.T + np.eye(3)
        P = 0.5 * (P + P.T)
        imm = B @ B.T / 3 + 0.5 * np.eye(3)
        imm = 0.5 * (imm + imm.T)
        mu, q0 = r.normal(size=3), r.normal(size=3)
        res = nuts_transition(seed, q0, mu, P, imm, eps, max_exp)
        cases.append(dict(name=name, sampler="nuts", seed=seed, eps=eps, max_exp=max_exp, mu=mu.tolist(), prec=P.tolist(),
                          imm=imm.tolist(), q0=q0.tolist(), expect=tolist(res)))
    A, B = r.normal(size=(3, 3)), r.normal(size=(3, 3))
    P = 0.5 * ((A @ A.T + np.eye(3)) + (A @ A.T + np.eye(3)).T)
    imm = B @ B.T / 3 + 0.5 * np.eye(3)
    imm = 0.5 * (imm + imm.T)
    mu, q0 = r.normal(size=3), r.normal(size=3)
    res = hmc_transition(11, q0, mu, P, imm, 0.35, 12)
    cases.append(dict(name="hmc-dense", sampler="hmc", seed=11, eps=0.35, L=12, mu=mu.tolist(), prec=P.tolist(),
                      imm=imm.tolist(), q0=q0.tolist(), expect=tolist(res)))
    doc = dict(note="Independent derivation (tests/golden/make_dense_pin.py, no oracle import) of dense-metric transitions, "
                    "D = 3, scheme-A RNG; NOT reference-generated (Aesara is absent here): a second restatement written from "
                    "SURVEY.md Appendix A with the literals of metrics.py:52-59",
               cases=cases)
    json.dump(doc, open(os.path.join(HERE, "dense_pin_v1.json"), "w"), indent=1)
    for c in cases:
        e = c["expect"]
        print(c["name"], e["position"], e["acceptance_probability"], e.get("n_leapfrog"), e.get("num_doublings"),
              e.get("is_turning"), [(t["direction"], t["length"]) for t in e.get("trace", [])])


if __name__ == "__main__":
    main()
