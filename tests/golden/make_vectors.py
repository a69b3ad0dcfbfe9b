#!/usr/bin/env python3
"""Generate tests/golden/vectors_v1.json.

The reference (aesara-devs/aehmc) cannot be imported in this container (Aesara / aeppl are
absent), so two kinds of vectors are committed:
  * "published": values the reference itself publishes (README.md:53-54, notebook cells
    :188 and :293-297, unit-test tables) -- copied as data, with their source line;
  * "derived": input/output vectors produced by THIS repo's C restatement (oracle/c), which
    is pinned to the published values above.  They are NOT reference-generated.
Run from the repo root:  python tests/golden/make_vectors.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402


def case(name, sampler, D, C, metric_kind, target_kind, eps, n_transitions, seed0, L=0, max_exp=6):
    import zlib
    r = np.random.default_rng(zlib.crc32(name.encode()))
    mu, sigma = r.normal(size=D).round(6), (0.5 + r.random(D)).round(6)
    prec = None
    if target_kind == "std_normal":
        otgt = co.Target(co.T_STD_NORMAL, D)
    elif target_kind == "iso":
        otgt = co.Target(co.T_ISO_GAUSSIAN, D)
    elif target_kind == "dense":  # dense MVN: precision = symmetrised inverse of A A^T / D + I, rounded so that
        A = r.normal(size=(D, D))  # the JSON holds exactly the matrix that was used
        prec = np.linalg.inv(A @ A.T / D + np.eye(D))
        prec = (0.5 * (prec + prec.T)).round(9)
        otgt = co.Target(co.T_DENSE_MVN, D, mu=mu, prec=prec)
    else:
        otgt = co.Target(co.T_DIAG_GAUSSIAN, D, mu=mu, sigma=sigma)
    if metric_kind == "scalar":
        imm = np.float64(0.7)
    elif metric_kind == "dense":
        B = r.normal(size=(D, D))
        imm = B @ B.T / D + np.eye(D)
        imm = (0.5 * (imm + imm.T)).round(9)
    else:
        imm = (0.5 + r.random(D)).round(6)
    metric = co.Metric(imm, D)
    seeds = [seed0 + c for c in range(C)]
    q0 = r.normal(size=(C, D)).round(6)
    q, U, g = co.new_state(otgt, q0.copy())
    rng = co.site_states(seeds, 4 if sampler == "nuts" else 2)
    steps = []
    for _ in range(n_transitions):
        if sampler == "nuts":
            res = co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp)
        else:
            res = co.hmc_step(otgt, metric, rng, eps, L, q, U, g)
        steps.append(dict(position=q.tolist(), potential_energy=U.tolist(),
                          acceptance_probability=res["acceptance_probability"].tolist(),
                          is_diverging=res["is_diverging"].astype(int).tolist(),
                          n_leapfrog=res["n_leapfrog"].tolist(),
                          num_doublings=res.get("num_doublings", np.zeros(C, int)).tolist(),
                          is_turning=res.get("is_turning", np.zeros(C, bool)).astype(int).tolist()))
    return dict(name=name, sampler=sampler, D=D, C=C, metric_kind=metric_kind, target_kind=target_kind,
                mu=mu.tolist(), sigma=sigma.tolist(), imm=np.atleast_1d(imm).tolist(), eps=eps, L=L,
                prec=None if prec is None else prec.tolist(),
                max_exp=max_exp, seeds=seeds, q0=q0.tolist(), steps=steps)


published = {
    "G1": {"source": "README.md:22-54", "seed": 0, "step_size": 1e-2, "inverse_mass_matrix": 1.0,
           "initial_position": 0.0, "position": 1.1034719409361107},
    "G2": {"source": "examples/LinearRegression.ipynb:293-297 (input :306)", "seed": 0, "step_size": 5e-5,
           "num_integration_steps": 1024, "initial_position": [3.0, float(np.log(0.21))],
           "position": [2.99946192, -1.30494977], "potential_energy": 12433.00653542,
           "potential_energy_grad": [-489.93218536, -22571.36970197], "p_accept": 1.0, "divergent": False},
    "G3": {"source": "examples/LinearRegression.ipynb:188", "position": [3.0, float(np.log(10.0))],
           "logprob": -32238.026021294307},
    "storage_indices": {"source": "tests/test_termination.py:51-62",
                        "table": {"0": [1, 0], "6": [3, 2], "7": [0, 2], "13": [2, 2], "15": [0, 3]}},
    "expansion_outcomes": {"source": "tests/test_trajectory.py:144-208",
                           "table": [[100000.0, True, False, 1], [1e-7, False, False, 10], [1.0, False, True, 1]]},
}
derived = [
    case("nuts_scalar_d1", "nuts", 1, 3, "scalar", "std_normal", 0.3, 3, 10),
    case("nuts_diag_d2", "nuts", 2, 4, "diag", "diag", 0.25, 3, 20),
    case("nuts_diag_d100", "nuts", 100, 3, "diag", "diag", 0.15, 2, 30),
    case("hmc_diag_d2", "hmc", 2, 4, "diag", "diag", 0.2, 3, 40, L=9),
    case("hmc_iso_d100", "hmc", 100, 3, "diag", "iso", 0.1, 2, 50, L=32),
]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_v1.json")
json.dump({"published": published, "derived": derived,
           "note": "derived vectors come from oracle/c (this repo), not from the reference"},
          open(out, "w"))
print("wrote", out, os.path.getsize(out), "bytes")
# v2 (round 2): the dense branch (dense inverse mass matrix and / or dense precision: no reference value
# exists for it anywhere) and a chain wide enough for the workgroup-per-chain kernels
derived2 = [
    case("nuts_dense_metric_d40", "nuts", 40, 3, "dense", "diag", 0.2, 2, 60, max_exp=7),
    case("nuts_dense_both_d40", "nuts", 40, 2, "dense", "dense", 0.2, 2, 70, max_exp=7),
    case("hmc_dense_both_d40", "hmc", 40, 2, "dense", "dense", 0.15, 2, 80, L=11),
    case("nuts_diag_d700", "nuts", 700, 2, "diag", "diag", 0.1, 2, 90, max_exp=8),
    case("hmc_diag_d1300", "hmc", 1300, 2, "diag", "diag", 0.08, 1, 95, L=16),
]
out2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_v2.json")
json.dump({"derived": derived2,
           "note": "derived vectors come from oracle/c (this repo), not from the reference; the dense branch is unpinned "
                   "by the reference (no published value exists)"}, open(out2, "w"))
print("wrote", out2, os.path.getsize(out2), "bytes")


# v3 (round 2): the regression target (the rows are regenerated from `data_seed`, as the notebook's own cell does)
# and a scalar-sized chain set large enough to take the ziggurat's redraw path several times
def linreg_case(name, sampler, N, C, eps, n_transitions, seed0, data_seed, L=0, max_exp=10):
    r = np.random.default_rng(data_seed)
    X = r.normal(0, 1, size=N)
    y = 3 * X + 0.5 * r.normal(0, 1, size=N)
    otgt = co.Target(co.T_LINREG, 2, X=X, y=y)
    imm = np.array([0.25 / N, 0.5 / N]).round(12)
    metric = co.Metric(imm, 2)
    seeds = [seed0 + c for c in range(C)]
    q0 = (np.array([3.0, np.log(0.5)]) + (0.5 / np.sqrt(N)) * np.random.default_rng(data_seed + 1).normal(size=(C, 2))).round(9)
    q, U, g = co.new_state(otgt, q0.copy())
    rng = co.site_states(seeds, 4 if sampler == "nuts" else 2)
    steps = []
    for _ in range(n_transitions):
        res = (co.nuts_step(otgt, metric, rng, eps, q, U, g, max_exp=max_exp) if sampler == "nuts"
               else co.hmc_step(otgt, metric, rng, eps, L, q, U, g))
        steps.append(dict(position=q.tolist(), potential_energy=U.tolist(),
                          acceptance_probability=res["acceptance_probability"].tolist(),
                          is_diverging=res["is_diverging"].astype(int).tolist(), n_leapfrog=res["n_leapfrog"].tolist(),
                          num_doublings=res.get("num_doublings", np.zeros(C, int)).tolist(),
                          is_turning=res.get("is_turning", np.zeros(C, bool)).astype(int).tolist()))
    return dict(name=name, sampler=sampler, D=2, C=C, metric_kind="diag", target_kind="linreg", N=N, data_seed=data_seed,
                imm=imm.tolist(), eps=eps, L=L, max_exp=max_exp, seeds=seeds, q0=q0.tolist(), steps=steps,
                mu=[], sigma=[], prec=None)


derived3 = [
    linreg_case("nuts_linreg_n1500", "nuts", 1500, 6, 0.5, 3, 100, 7),
    linreg_case("nuts_linreg_n12000", "nuts", 12000, 5, 0.5, 2, 110, 8),
    linreg_case("hmc_linreg_n1500", "hmc", 1500, 6, 0.3, 2, 120, 9, L=12),
    case("nuts_d1_40_chains", "nuts", 1, 40, "diag", "diag", 0.3, 4, 130),
]
out3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_v3.json")
json.dump({"derived": derived3,
           "note": "derived vectors come from oracle/c (this repo), not from the reference; regression rows: "
                   "r = default_rng(data_seed); X = r.normal(0, 1, N); y = 3 X + 0.5 r.normal(0, 1, N)"}, open(out3, "w"))
print("wrote", out3, os.path.getsize(out3), "bytes")
