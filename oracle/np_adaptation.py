"""CPU restatement (numpy) of aehmc's warm-up: dual averaging, Welford, mass-matrix and
window adaptation.  TEST INFRASTRUCTURE ONLY (same rules as np_oracle.py).

Follows /root/reference/aehmc/algorithms.py, step_size.py, mass_matrix.py and
window_adaptation.py literally, one chain at a time, including the quirks listed in
SURVEY.md 8f-2 (DA `init(mu)` starts the iterate at 0 so the first step size is
exp(0) = 1 and `mu = initial_step_size`, not its log; the averaged iterate uses the
PREVIOUS iterate; a slow window's end re-inits DA with mu = exp(x)).
Pinned by the known-answer tests of the reference (tests/test_adaptation.py:9-22,
tests/test_algorithms.py) in tests/test_adaptation_oracle.py.
"""
from __future__ import annotations

from typing import List, NamedTuple, Tuple

import numpy as np


# ------------------------------------------------------------------ algorithms.py:9-117
class DualAveragingState(NamedTuple):
    step: int
    iterates: float
    iterates_avg: float
    gradient_avg: float
    shrinkage_pts: float


def dual_averaging(gamma: float = 0.05, t0: int = 10, kappa: float = 0.75):
    def init(mu):  # algorithms.py:56-76
        return DualAveragingState(1, 0.0, 0.0, 0.0, mu)

    def update(gradient, state):  # algorithms.py:78-115
        eta = 1.0 / (state.step + t0)
        new_gradient_avg = (1.0 - eta) * state.gradient_avg + eta * gradient
        new_x = state.shrinkage_pts - (np.sqrt(state.step) / gamma) * new_gradient_avg
        x_eta = float(state.step) ** (-kappa)
        new_x_avg = x_eta * state.iterates + (1.0 - x_eta) * state.iterates_avg
        return state._replace(step=state.step + 1, iterates=float(new_x),
                              iterates_avg=float(new_x_avg), gradient_avg=float(new_gradient_avg))

    return init, update


# ------------------------------------------------------------------ algorithms.py:120-204
def welford_covariance(compute_covariance: bool):
    def init(n_dims: int):
        if n_dims == 0:
            return np.float64(0.0), np.float64(0.0), 0
        mean = np.zeros(n_dims)
        m2 = np.zeros((n_dims, n_dims)) if compute_covariance else np.zeros(n_dims)
        return mean, m2, 0

    def update(value, mean, m2, sample_size):
        sample_size = sample_size + 1
        delta = value - mean
        mean = mean + delta / sample_size
        updated_delta = value - mean
        if compute_covariance and np.ndim(mean) > 0:
            m2 = m2 + np.outer(updated_delta, delta)
        else:
            m2 = m2 + updated_delta * delta
        return mean, m2, sample_size

    def final(m2, sample_size):
        with np.errstate(divide="ignore", invalid="ignore"):
            return m2 / (sample_size - 1)

    return init, update, final


# ------------------------------------------------------------------ step_size.py:9-100
def dual_averaging_adaptation(target_acceptance_rate=0.8, gamma=0.05, t0=10, kappa=0.75):
    da_init, da_update = dual_averaging(gamma, t0, kappa)

    def update(acceptance_probability, state):
        return da_update(target_acceptance_rate - acceptance_probability, state)

    return da_init, update


# ------------------------------------------------------------------ mass_matrix.py:12-120
def covariance_adaptation(is_mass_matrix_full: bool = False):
    wc_init, wc_update, wc_final = welford_covariance(is_mass_matrix_full)

    def init(n_dims):
        if n_dims == 0:
            imm = np.float64(1.0)
        elif is_mass_matrix_full:
            imm = np.eye(n_dims)
        else:
            imm = np.ones(n_dims)
        return imm, wc_init(n_dims)

    def update(position, wc_state):
        return wc_update(position, *wc_state)

    def final(wc_state):
        _, m2, n = wc_state
        covariance = wc_final(m2, n)
        scaled = (n / (n + 5)) * covariance
        shrinkage = 1e-3 * (5 / (n + 5))
        if np.ndim(covariance) > 0 and is_mass_matrix_full:
            return scaled + shrinkage * np.eye(covariance.shape[0])
        return scaled + shrinkage

    return init, update, final


# ------------------------------------------------------------------ window_adaptation.py:230-327
def build_schedule(num_steps, initial_buffer_size=75, final_buffer_size=50,
                   first_window_size=25) -> List[Tuple[int, bool]]:
    schedule: List[Tuple[int, bool]] = []
    if num_steps < 20:
        return [(0, False)] * num_steps
    if initial_buffer_size + first_window_size + final_buffer_size > num_steps:
        initial_buffer_size = int(0.15 * num_steps)
        final_buffer_size = int(0.1 * num_steps)
        first_window_size = num_steps - initial_buffer_size - final_buffer_size
    schedule += [(0, False)] * initial_buffer_size
    final_buffer_start = num_steps - final_buffer_size
    next_size, next_start = first_window_size, initial_buffer_size
    while next_start < final_buffer_start:
        cur_start, cur_size = next_start, next_size
        if 3 * cur_size <= final_buffer_start - cur_start:
            next_size = 2 * cur_size
        else:
            cur_size = final_buffer_start - cur_start
        next_start = cur_start + cur_size
        schedule += [(1, False)] * (next_start - 1 - cur_start)
        schedule.append((1, True))
    schedule += [(0, False)] * (num_steps - final_buffer_start)
    return schedule


# ------------------------------------------------------------------ window_adaptation.py:119-227, 17-116
def window_adaptation(num_steps, is_mass_matrix_full=False, initial_step_size=1.0,
                      target_acceptance_rate=0.80):
    mm_init, mm_update, mm_final = covariance_adaptation(is_mass_matrix_full)
    da_init, da_update = dual_averaging_adaptation(target_acceptance_rate)
    schedule = build_schedule(num_steps)

    def init(position):
        num_dims = 0 if np.ndim(position) == 0 else np.shape(position)[0]
        imm, mm_state = mm_init(num_dims)
        da_state = da_init(initial_step_size)
        return (da_state, mm_state), (float(np.exp(da_state.iterates)), imm)

    def update(step, warmup_state, parameters, position, p_accept):
        da_state, mm_state = warmup_state
        _, imm = parameters
        stage, is_middle_window_end = schedule[step]
        da_state = da_update(p_accept, da_state)          # fast and slow
        if stage != 0:
            mm_state = mm_update(position, mm_state)      # slow only
        step_size = float(np.exp(da_state.iterates))
        if is_middle_window_end:                          # slow_final :165-182
            imm = mm_final(mm_state)
            num_dims = 0 if np.ndim(imm) == 0 else np.shape(imm)[0]
            _, mm_state = mm_init(num_dims)
            step_size = float(np.exp(da_state.iterates))
            da_state = da_init(step_size)
        if step == num_steps - 1:                         # final :184-190
            step_size = float(np.exp(da_state.iterates_avg))
        return (da_state, mm_state), (step_size, imm)

    return init, update


def run(kernel, initial_state, num_steps=1000, *, is_mass_matrix_full=False,
        initial_step_size=1.0, target_acceptance_rate=0.80):
    """window_adaptation.run :17-116.  `kernel(state, step_size, imm) -> Diagnostics`
    (any object with .state and .acceptance_probability)."""
    init, update = window_adaptation(num_steps, is_mass_matrix_full, initial_step_size,
                                     target_acceptance_rate)
    warmup_state, parameters = init(initial_state.position)
    state = initial_state
    for step in range(num_steps):
        info = kernel(state, *parameters)
        state = info.state._replace(momentum=None)
        warmup_state, parameters = update(step, warmup_state, parameters, info.state.position,
                                          info.acceptance_probability)
    return state, parameters
