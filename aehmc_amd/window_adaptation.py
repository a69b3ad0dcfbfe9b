"""Stan-style window adaptation of the step size and the inverse mass matrix (diagonal, or
dense per chain with ``is_mass_matrix_full``), one adaptation per chain (reference: aehmc/window_adaptation.py, step_size.py,
mass_matrix.py, algorithms.py).  The schedule is host logic; the per-chain dual-averaging /
Welford updates run in one HIP kernel per warm-up step (`aehmc_adapt_update`)."""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional, Tuple

import torch

from ._common import Layout
from .engine import PerChain, _dev_f64, get_engine
from .integrators import IntegratorState
from .step_size import DualAveragingState


def build_schedule(num_steps: int, initial_buffer_size: int = 75, final_buffer_size: int = 50,
                   first_window_size: int = 25) -> List[Tuple[int, bool]]:
    """(window_label, is_middle_window_end) per warm-up step (reference:
    aehmc/window_adaptation.py:230-327): a fast initial buffer, slow windows doubling in
    size with no memory, a fast final buffer; labels 0 = fast, 1 = slow."""
    if num_steps < 20:  # too short for mass-matrix adaptation
        return [(0, False)] * num_steps
    if initial_buffer_size + first_window_size + final_buffer_size > num_steps:
        initial_buffer_size = int(0.15 * num_steps)
        final_buffer_size = int(0.1 * num_steps)
        first_window_size = num_steps - initial_buffer_size - final_buffer_size
    slow_end = num_steps - final_buffer_size
    labels = [(0, False)] * initial_buffer_size
    start, size = initial_buffer_size, first_window_size
    while start < slow_end:
        if 3 * size <= slow_end - start:
            this, size = size, 2 * size
        else:
            this = slow_end - start
        labels += [(1, False)] * (this - 1) + [(1, True)]
        start += this
    return labels + [(0, False)] * (num_steps - slow_end)


def run(kernel, initial_state: IntegratorState, num_steps=1000, *, is_mass_matrix_full=False,
        initial_step_size=1.0, target_acceptance_rate=0.80, fused=True, num_integration_steps=None
        ) -> Tuple[IntegratorState, Tuple, Dict]:
    """Warm a kernel up for ``num_steps`` transitions (reference:
    aehmc/window_adaptation.py:17-116).  Returns ``(last_chain_state, (step_size,
    inverse_mass_matrix), updates)`` where the parameters are ``PerChain`` values -- one
    step size and one (diagonal or dense) inverse mass matrix per chain, exactly as running the
    reference once per chain would produce -- to be passed back to ``kernel``.

    With a NUTS or HMC kernel of this package the whole loop runs inside one C-ABI call
    (``aehmc_nuts_warmup`` / ``aehmc_hmc_warmup``: transition, adaptation update, transition, ... enqueued back to back);
    ``fused=False`` -- and any other kernel -- takes the step-by-step loop below, which issues the
    same kernels in the same order (identical results).

    An HMC kernel (``hmc.new_kernel``) takes a fourth argument; pass its fixed trajectory length as
    ``num_integration_steps`` and the loop calls ``kernel(state, step_size, imm, num_integration_steps)``
    (the reference's loop calls ``kernel(chain_state, *parameters)``, window_adaptation.py:66, so there an
    HMC kernel has to be wrapped in a lambda that closes over the length -- which works here too)."""
    if getattr(kernel, "_hmc", None) is not None and num_integration_steps is None:
        raise ValueError("window_adaptation.run with an HMC kernel needs num_integration_steps")
    extra = () if num_integration_steps is None else (int(num_integration_steps),)
    eng = get_engine()
    pos = initial_state.position
    srng_chains = getattr(kernel, "num_chains", None)
    batched = getattr(kernel, "batched", pos.ndim == 2)
    layout = Layout(tuple(pos.shape), batched, srng_chains or (pos.shape[0] if batched else 1))
    C, D = layout.C, layout.D
    scalar_position = (len(layout.user_shape) - (1 if batched else 0)) == 0
    full = bool(is_mass_matrix_full) and not scalar_position  # mass_matrix.py:54-57: a scalar stays a scalar
    if full and D > 2048:  # one wavefront factors each chain's matrix (tests: up to D = 1024)
        raise ValueError("is_mass_matrix_full keeps one dense D x D matrix per chain (as the reference does) and "
                         "is supported up to D = 2048")
    st, cst = eng.adapt_alloc(C, D, full)
    eng.adapt_init(C, D, float(initial_step_size), cst)
    schedule = build_schedule(int(num_steps))

    def imm_param():
        return PerChain(st["imm"].reshape(C) if scalar_position else st["imm"], st["sqrt_mass"])

    nk = getattr(kernel, "_nuts", None)
    if fused and nk is not None and len(schedule) > 0:
        from ._common import diagnostics, state_rows
        from .engine import rng_to_device
        if "rng" not in nk["holder"] or nk["holder"]["rng"].device != eng.device:  # (uploaded at construction, maybe elsewhere)
            nk["holder"]["rng"] = (nk["holder"]["rng"].to(eng.device) if "rng" in nk["holder"]
                                   else rng_to_device(nk["rng_host"], eng.device))
        q, U, g = state_rows(initial_state, layout, eng.device)
        eng.set_target(nk["logprob_fn"], D)
        out = eng.nuts_warmup(nk["holder"]["rng"], schedule, float(target_acceptance_rate),
                              nk["max_num_expansions"], nk["divergence_threshold"], q, U, g, st, cst, imm_param())
        info = diagnostics(layout, q, U, g, out, True)
        state, updates = info.state._replace(momentum=None), {nk["srng"]: nk["holder"]["rng"]}
        schedule = []
    elif fused and getattr(kernel, "_hmc", None) is not None and len(schedule) > 0:
        from ._common import diagnostics, state_rows
        from .engine import rng_to_device
        hk = kernel._hmc
        if "rng" not in hk["holder"] or hk["holder"]["rng"].device != eng.device:
            hk["holder"]["rng"] = (hk["holder"]["rng"].to(eng.device) if "rng" in hk["holder"]
                                   else rng_to_device(hk["rng_host"], eng.device))
        q, U, g = state_rows(initial_state, layout, eng.device)
        eng.set_target(hk["logprob_fn"], D)
        out = eng.hmc_warmup(hk["holder"]["rng"], schedule, float(target_acceptance_rate), extra[0],
                             hk["divergence_threshold"], q, U, g, st, cst, imm_param())
        info = diagnostics(layout, q, U, g, out, False)
        state, updates = info.state._replace(momentum=None), {hk["srng"]: hk["holder"]["rng"]}
        schedule = []
    else:
        state, updates = initial_state, {}
    for i, (stage, window_end) in enumerate(schedule):
        info, updates = kernel(state, PerChain(st["step_size"]), imm_param(), *extra)
        state = info.state._replace(momentum=None)
        eng.adapt_update(C, D, stage, window_end, i == len(schedule) - 1, float(target_acceptance_rate),
                         info.acceptance_probability.reshape(C).contiguous(),
                         state.position.reshape(C, D).contiguous(), cst)
    step_size = st["step_size"].clone()
    imm, sqrt_mass = st["imm"].clone(), st["sqrt_mass"].clone()
    return state, (PerChain(layout.per_chain(step_size)),
                   PerChain(imm.reshape(C) if scalar_position else imm, sqrt_mass)), updates


class WarmupState(NamedTuple):
    """window_adaptation.py:119-227's ``warmup_state = (da_state, mm_state)`` (DualAveragingState,
    algorithms.py:9-14; Welford state ``(mean, m2, sample_size)``, algorithms.py:141-165) plus the current
    parameters' device arrays the update kernel rewrites (step size, inverse mass matrix, its square root)."""
    da_state: DualAveragingState
    mm_state: Tuple
    step_size: torch.Tensor
    imm: torch.Tensor
    sqrt_mass: torch.Tensor
    work: Optional[torch.Tensor] = None
    position_shape: Tuple = ()   # user-facing shape of the chain position and whether it has a leading chain axis
    batched: bool = False


def window_adaptation(num_steps: int, is_mass_matrix_full: bool = False, initial_step_size=1.0,
                      target_acceptance_rate=0.80):
    """The warm-up as ``(init, update)`` for callers that drive the loop themselves (reference:
    aehmc/window_adaptation.py:119-227 -- ``run`` above is that loop in one engine call):

        init, update = window_adaptation(num_steps)
        warmup_state, parameters = init(state)              # parameters = (step_size, inverse_mass_matrix)
        for i in range(num_steps):
            info, _ = kernel(state, *parameters)
            state = info.state._replace(momentum=None)
            warmup_state, parameters = update(i, warmup_state, parameters, info)

    ``update`` is one launch of the warm-up kernel (``aehmc_adapt_update``: dual averaging in every stage, Welford in
    the slow windows, new metric + restart at a window end, the averaged step size after the last step) on COPIES of
    the state arrays -- states are values, as in the reference; the parameters are ``PerChain`` values (one
    adaptation per chain)."""
    schedule = build_schedule(int(num_steps))

    def _layout(position, num_chains):
        shape = tuple(position.shape)
        batched = num_chains is not None or len(shape) == 2
        C = (num_chains if num_chains is not None else shape[0]) if batched else 1
        layout = Layout(shape, batched, C)
        scalar_position = (len(shape) - (1 if batched else 0)) == 0
        return layout, scalar_position

    def _params(layout, scalar_position, ws: WarmupState):
        C = layout.C
        return (PerChain(layout.per_chain(ws.step_size)),
                PerChain(ws.imm.reshape(C) if scalar_position else ws.imm, ws.sqrt_mass))

    def _cstate(eng, ws: WarmupState, full):
        from . import _lib
        da, (mean, m2, n) = ws.da_state, ws.mm_state
        ptr = dict(da_step=da.step, da_x=da.iterates, da_x_avg=da.iterates_avg, da_g_avg=da.gradient_avg,
                   da_mu=da.shrinkage_pts, wc_mean=mean, wc_m2=m2, wc_n=n, step_size=ws.step_size, imm=ws.imm,
                   sqrt_mass=ws.sqrt_mass)
        if ws.work is not None:
            ptr["work"] = ws.work
        return _lib.CAdaptState(full=int(bool(full)), **{k: v.data_ptr() for k, v in ptr.items()})

    def init(initial_chain_state: IntegratorState, num_chains: Optional[int] = None):
        """window_adaptation.py:130-143: identity metric, dual averaging started at ``initial_step_size`` (so the
        first step size is exp(0) = 1, algorithms.py:56-76)."""
        eng = get_engine()
        layout, scalar_position = _layout(initial_chain_state.position, num_chains)
        C, D = layout.C, layout.D
        full = bool(is_mass_matrix_full) and not scalar_position
        st, cst = eng.adapt_alloc(C, D, full)
        eng.adapt_init(C, D, float(initial_step_size), cst)
        ws = WarmupState(DualAveragingState(st["da_step"], st["da_x"], st["da_x_avg"], st["da_g_avg"], st["da_mu"]),
                         (st["wc_mean"], st["wc_m2"], st["wc_n"]), st["step_size"], st["imm"], st["sqrt_mass"],
                         st.get("work"), layout.user_shape, layout.C > 1 or num_chains is not None or
                         len(layout.user_shape) == 2)
        return ws, _params(layout, scalar_position, ws)

    def update(step: int, warmup_state: WarmupState, parameters, chain_state):
        """window_adaptation.py:192-214 for warm-up step ``step`` (0-based) after the transition ``chain_state``."""
        del parameters  # (the arrays in warmup_state ARE the current parameters)
        eng = get_engine()
        position = chain_state.state.position
        C = warmup_state.step_size.numel()
        if tuple(position.shape) != tuple(warmup_state.position_shape):
            raise ValueError(f"position has shape {tuple(position.shape)}, the warm-up was initialised with "
                             f"{tuple(warmup_state.position_shape)}")
        layout, scalar_position = _layout(position, C if warmup_state.batched else None)
        D = layout.D
        full = warmup_state.imm.ndim == 3
        da, (mean, m2, n) = warmup_state.da_state, warmup_state.mm_state
        ws = WarmupState(DualAveragingState(*(t.clone() for t in da)), (mean.clone(), m2.clone(), n.clone()),
                         warmup_state.step_size.clone(), warmup_state.imm.clone(), warmup_state.sqrt_mass.clone(),
                         warmup_state.work, warmup_state.position_shape, warmup_state.batched)
        stage, window_end = schedule[int(step)]
        eng.adapt_update(C, D, stage, window_end, int(step) == len(schedule) - 1, float(target_acceptance_rate),
                         _dev_f64(chain_state.acceptance_probability, eng.device).reshape(C).contiguous(),
                         _dev_f64(position, eng.device).reshape(C, D).contiguous(), _cstate(eng, ws, full))
        return ws, _params(layout, scalar_position, ws)

    return init, update
