"""CPU restatement (numpy) of aehmc's HMC / NUTS transition.  TEST INFRASTRUCTURE ONLY.

This module is the *parity oracle*: a literal, single-chain, eager transcription of
what the reference's symbolic Aesara graphs compute.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg --
never by the product package ``aehmc_amd`` (whose hot path is HIP only).

Pinned against the values the reference publishes (see tests/test_oracle_golden.py):
  G1  README.md:22-54   NUTS seed 0, N(0,1) -> 1.1034719409361107 (bit-exact)
  G2  examples/LinearRegression.ipynb:293-297  HMC single step
  G3  examples/LinearRegression.ipynb:188      regression log-density
  and the known-answer tables of tests/test_termination.py, tests/test_metrics.py,
  tests/test_trajectory.py, tests/test_integrators.py.

Every function cites the reference file:line (relative to /root/reference) it follows.

PARITY UNPINNED for two things the reference publishes no value for: (1) dense-metric
trajectories (only the exact kinetic-energy / is_turning unit cases of tests/test_metrics.py
exist), (2) RNG consumption after a sub-trajectory whose first step diverged
(trajectory.py:336: we assume the inner scan still executes -- its RNG updates are outputs
of the compiled function, so a lazy ifelse cannot skip it).  For those, parity means
"HIP == this restatement"; for (1) there is in addition an exact invariance that ties the dense
branch to the pinned diagonal one: a whole transition is equivariant under q' = A q for
lower-triangular A (tests/test_oracle_golden.py::
test_dense_branch_equals_diagonal_branch_under_triangular_map).

Third-party arithmetic that is NOT in /root/reference: aesara>=2.8.11 / aeppl>=0.1.4
(pyproject.toml:18-19, lower bounds only).  Their RNG is restated as "scheme A":
``RandomStream(seed)`` keeps ``SeedSequence(seed)``; each ``srng.<dist>()`` call site,
in graph-construction order, owns ``default_rng(seedseq.spawn(1)[0])``;
normal -> ``Generator.normal``; bernoulli -> ``Generator.binomial(1, p)``.
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional, Tuple

import numpy as np

LOG_SQRT_2PI = float(np.log(np.sqrt(2.0 * np.pi)))  # aeppl normal logprob constant


# --------------------------------------------------------------------------------------
# RNG scheme A
# --------------------------------------------------------------------------------------
class RandomStream:
    """Look-alike of aesara.tensor.random.utils.RandomStream (scheme A)."""

    def __init__(self, seed: int):
        self.seed_seq = np.random.SeedSequence(seed)

    def site(self) -> np.random.Generator:
        """One RNG call site == one spawned child generator (creation order matters)."""
        return np.random.default_rng(self.seed_seq.spawn(1)[0])


def bernoulli(gen: np.random.Generator, p: float) -> bool:
    # aesara BernoulliRV -> scipy.stats.bernoulli.rvs -> Generator.binomial(1, p)
    return bool(gen.binomial(1, p))


# --------------------------------------------------------------------------------------
# Targets (stand-ins for the symbolic logprob_fn); potential = -logprob
# --------------------------------------------------------------------------------------
class StdNormal:
    """aeppl logprob of N(0,1) per coordinate (README.md:27-36):
    logp = -0.5*y**2 - log(sqrt(2*pi)); scalar or vector position."""

    kind = "std_normal"

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        if q.ndim == 0:
            return float(0.5 * (q * q) + LOG_SQRT_2PI), np.float64(q)
        u = 0.0
        for x in q:  # sequential sum (matches the C restatement)
            u += 0.5 * (x * x) + LOG_SQRT_2PI
        return float(u), q.copy()


class IsoGaussian:
    """U = 0.5*||q||^2 (SURVEY 8d c2 synthetic target)."""

    kind = "iso_gaussian"

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        if q.ndim == 0:
            return float(0.5 * (q * q)), np.float64(q)
        u = 0.0
        for x in q:
            u += x * x
        return float(0.5 * u), q.copy()


class DiagGaussian:
    """N(mu, diag(sigma^2)); U = sum 0.5*((q-mu)/sigma)^2 + log(sigma) + log(sqrt(2pi)).
    Gradient: (q-mu)/sigma/sigma."""

    kind = "diag_gaussian"

    def __init__(self, mu, sigma):
        self.mu = np.atleast_1d(np.asarray(mu, dtype=np.float64))
        self.sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float64))

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        scalar = q.ndim == 0
        qq = np.atleast_1d(q)
        z = (qq - self.mu) / self.sigma
        u = 0.0
        for zi, si in zip(z, self.sigma):
            u += 0.5 * (zi * zi) + np.log(si) + LOG_SQRT_2PI
        g = z / self.sigma
        if scalar:
            return float(u), np.float64(g[0])
        return float(u), g


class DenseMVN:
    """U = 0.5*(q-mu)^T P (q-mu), grad = P (q-mu); P symmetric precision."""

    kind = "dense_mvn"

    def __init__(self, mu, precision):
        self.mu = np.asarray(mu, dtype=np.float64)
        self.P = np.asarray(precision, dtype=np.float64)

    def __call__(self, q):
        r = np.asarray(q, dtype=np.float64) - self.mu
        g = self.P @ r
        return float(0.5 * np.dot(r, g)), g


class LinearRegression:
    """examples/LinearRegression.ipynb:126-166: w~N(0,1), n~Gamma(2,1), y_i~N(X_i w, n),
    sampled in q=[w, log n] (log transform, log-Jacobian +log n)."""

    kind = "linear_regression"

    def __init__(self, X, y):
        self.X = np.asarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)

    def logp(self, q):
        w, ell = float(q[0]), float(q[1])
        n = np.exp(ell)
        r = self.y - self.X * w
        N = self.X.shape[0]
        lp_w = -0.5 * w * w - LOG_SQRT_2PI
        lp_n = np.log(n) - n + ell  # Gamma(2,1): log n - n - lgamma(2); + Jacobian
        lp_y = -0.5 * (np.sum(r * r) / (n * n)) - N * LOG_SQRT_2PI - N * ell
        return float(lp_w + lp_n + lp_y)

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        w, ell = float(q[0]), float(q[1])
        n = np.exp(ell)
        r = self.y - self.X * w
        N = self.X.shape[0]
        n2 = n * n
        s_xr = float(np.sum(self.X * r))
        s_rr = float(np.sum(r * r))
        dw = -w + s_xr / n2
        dl = 2.0 - n - N + s_rr / n2
        return -self.logp(q), np.array([-dw, -dl])


# --------------------------------------------------------------------------------------
# a1  IntegratorState  (integrators.py:7-11)
# --------------------------------------------------------------------------------------
class IntegratorState(NamedTuple):
    position: np.ndarray
    momentum: Optional[np.ndarray]
    potential_energy: float
    potential_energy_grad: np.ndarray


def new_state(q, target) -> IntegratorState:
    """a2: hmc.py:16-40."""
    u, g = target(q)
    return IntegratorState(np.asarray(q, dtype=np.float64), None, u, g)


# --------------------------------------------------------------------------------------
# a4-a6  gaussian_metric  (metrics.py:10-106)
# --------------------------------------------------------------------------------------
def gaussian_metric(inverse_mass_matrix):
    imm = np.asarray(inverse_mass_matrix, dtype=np.float64)
    if imm.ndim == 0:  # metrics.py:44-47
        shape: Tuple = ()
        mass_matrix_sqrt = np.sqrt(np.reciprocal(imm))
        dot = lambda x, y: x * y
        matmul = lambda x, y: x * y
    elif imm.ndim == 1:  # metrics.py:48-51
        shape = (imm.shape[0],)
        mass_matrix_sqrt = np.sqrt(np.reciprocal(imm))
        dot = np.dot
        matmul = lambda x, y: x * y
    elif imm.ndim == 2:  # metrics.py:52-59
        import scipy.linalg

        shape = (imm.shape[0],)
        L = np.linalg.cholesky(imm)
        mass_matrix_sqrt = scipy.linalg.solve_triangular(
            L, np.eye(shape[0]), lower=True, trans=1
        )
        dot = np.dot
        matmul = np.dot
    else:  # metrics.py:60-63
        raise ValueError(
            f"Expected a mass matrix of dimension 1 (diagonal) or 2, got {imm.ndim}"
        )

    def momentum_generator(gen: np.random.Generator):
        z = gen.normal(0, 1, size=shape)  # metrics.py:66
        return matmul(mass_matrix_sqrt, z)

    def kinetic_energy(p):  # metrics.py:70-73
        velocity = matmul(imm, p)
        return 0.5 * dot(velocity, p)

    def velocity(p):  # d(kinetic_energy)/dp by autodiff == imm o p (imm symmetric)
        return matmul(imm, p)

    def is_turning(p_left, p_right, p_sum):  # metrics.py:75-104
        v_left = matmul(imm, p_left)
        v_right = matmul(imm, p_right)
        rho = p_sum - (p_right + p_left) / 2
        return bool((np.dot(v_left, rho) <= 0) | (np.dot(v_right, rho) <= 0))

    return momentum_generator, kinetic_energy, is_turning, velocity


# --------------------------------------------------------------------------------------
# a3  velocity_verlet.one_step  (integrators.py:54-73)
# --------------------------------------------------------------------------------------
def velocity_verlet(target, velocity_fn) -> Callable:
    a1, b1 = 0, 0.5
    a2 = 1 - 2 * a1

    def one_step(state: IntegratorState, step_size: float) -> IntegratorState:
        momentum = state.momentum - b1 * step_size * state.potential_energy_grad
        kinetic_grad = velocity_fn(momentum)
        position = state.position + a2 * step_size * kinetic_grad
        u, g = target(position)
        momentum = momentum - b1 * step_size * g
        return IntegratorState(position, momentum, u, g)

    return one_step


# --------------------------------------------------------------------------------------
# a7/a8/a21  HMC  (trajectory.py:31-107, hmc.py:43-204)
# --------------------------------------------------------------------------------------
class Diagnostics(NamedTuple):  # a22 trajectory.py:379-384 (+ n_leapfrog, build extension)
    state: IntegratorState
    acceptance_probability: float
    num_doublings: Optional[int]
    is_turning: Optional[bool]
    is_diverging: bool
    n_leapfrog: int = 0


def hmc_kernel(srng: RandomStream, target, divergence_threshold=1000):
    g_momentum = srng.site()  # site #1 hmc.py:122
    g_accept = srng.site()  # site #2 hmc.py:194

    def step(state, step_size, inverse_mass_matrix, num_integration_steps):
        momentum_generator, kinetic_energy, _, velocity = gaussian_metric(
            inverse_mass_matrix
        )
        integrator = velocity_verlet(target, velocity)
        state = state._replace(momentum=momentum_generator(g_momentum))
        new = state
        for _ in range(int(num_integration_steps)):  # trajectory.py:86-95
            new = integrator(new, step_size)
        new = new._replace(momentum=-1.0 * new.momentum)  # hmc.py:185
        energy = state.potential_energy + kinetic_energy(state.momentum)
        new_energy = new.potential_energy + kinetic_energy(new.momentum)
        delta = energy - new_energy
        if np.isnan(delta):
            delta = -np.inf  # hmc.py:190
        is_div = bool(abs(delta) > divergence_threshold)
        with np.errstate(over="ignore"):
            p_accept = float(np.clip(np.exp(delta), 0, 1.0))
        do_accept = bernoulli(g_accept, p_accept)
        final = new if do_accept else state
        return Diagnostics(final, p_accept, None, None, is_div, int(num_integration_steps))

    return step


# --------------------------------------------------------------------------------------
# a9-a12  proposals.py
# --------------------------------------------------------------------------------------
class ProposalState(NamedTuple):
    state: IntegratorState
    energy: float
    weight: float
    sum_log_p_accept: float


def proposal_generator(kinetic_energy, divergence_threshold):
    def update(initial_energy, state):  # proposals.py:19-62
        new_energy = state.potential_energy + kinetic_energy(state.momentum)
        delta = initial_energy - new_energy
        if np.isnan(delta):
            delta = -np.inf
        is_div = bool(abs(delta) > divergence_threshold)
        weight = delta
        log_p_accept = 0.0 if delta > 0 else delta
        return ProposalState(state, new_energy, weight, log_p_accept), is_div

    return update


def _expit(x):
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-x))


def maybe_update_proposal(do_accept, proposal, new_proposal):  # proposals.py:137-174
    w = float(np.logaddexp(proposal.weight, new_proposal.weight))
    s = float(np.logaddexp(proposal.sum_log_p_accept, new_proposal.sum_log_p_accept))
    src = new_proposal if do_accept else proposal
    return ProposalState(src.state, src.energy, w, s)


def progressive_uniform_sampling(gen, proposal, new_proposal):  # proposals.py:72-102
    with np.errstate(invalid="ignore"):
        p_accept = _expit(new_proposal.weight - proposal.weight)
    if np.isnan(p_accept):
        p_accept = 0.0
    return maybe_update_proposal(bernoulli(gen, p_accept), proposal, new_proposal)


def progressive_biased_sampling(gen, proposal, new_proposal):  # proposals.py:105-134
    with np.errstate(over="ignore", invalid="ignore"):
        p_accept = float(np.clip(np.exp(new_proposal.weight - proposal.weight), 0.0, 1.0))
    return maybe_update_proposal(bernoulli(gen, p_accept), proposal, new_proposal)


# --------------------------------------------------------------------------------------
# a13-a16  termination.py
# --------------------------------------------------------------------------------------
class TerminationState(NamedTuple):
    momentum_checkpoints: np.ndarray
    momentum_sum_checkpoints: np.ndarray
    min_index: int
    max_index: int


def find_storage_indices(step: int) -> Tuple[int, int]:
    """termination.py:192-235, literal loops (the two `scan`s with `until`)."""
    nc0, nc1 = step, -1
    for _ in range(step + 1):
        do_stop = (nc0 & 1) == 0
        nc0, nc1 = nc0 // 2, nc1 + 1
        if do_stop:
            break
    num_subtrees = nc1
    nc0, nc1 = step // 2, 0
    for _ in range(step + 1):
        do_stop = nc0 == 0
        nc0, nc1 = nc0 // 2, nc1 + (nc0 & 1)
        if do_stop:
            break
    idx_max = nc1
    return idx_max - num_subtrees + 1, idx_max


def iterative_uturn(is_turning_fn):
    def new_termination_state(position, max_num_doublings):  # termination.py:43-83
        position = np.asarray(position)
        if position.ndim == 0:
            shp = (max_num_doublings,)
        else:
            shp = (max_num_doublings, position.shape[0])
        return TerminationState(np.zeros(shp), np.zeros(shp), 0, 0)

    def update(state, momentum_sum, momentum, step):  # termination.py:85-131
        if step == 0:
            idx_min, idx_max = state.min_index, state.max_index  # stale (quirk 2)
        else:
            idx_min, idx_max = find_storage_indices(step)
        ck, cks = state.momentum_checkpoints, state.momentum_sum_checkpoints
        if step % 2 == 0:
            ck, cks = ck.copy(), cks.copy()
            ck[idx_max] = momentum
            cks[idx_max] = momentum_sum
        return TerminationState(ck, cks, idx_min, idx_max)

    def is_iterative_turning(state, momentum_sum, momentum):  # termination.py:133-187
        if state.max_index < state.min_index:
            return False
        i = state.max_index
        crit = False
        for _ in range(state.max_index + 2):
            sub = momentum_sum - state.momentum_sum_checkpoints[i] + state.momentum_checkpoints[i]
            crit = is_turning_fn(state.momentum_checkpoints[i], momentum, sub)
            reached = (i - 1) < state.min_index
            i = i - 1
            if crit or reached:
                break
        return bool(crit)

    return new_termination_state, update, is_iterative_turning


# --------------------------------------------------------------------------------------
# a17  dynamic_integration.integrate  (trajectory.py:154-374)
# --------------------------------------------------------------------------------------
def dynamic_integration(g_uniform, integrator, kinetic_energy, update_termination_state,
                        is_criterion_met, divergence_threshold, counter=None):
    generate_proposal = proposal_generator(kinetic_energy, divergence_threshold)

    def integrate(previous_last_state, direction, termination_state, max_num_steps,
                  step_size, initial_energy):
        # first step, outside the loop: trajectory.py:276-305
        state = integrator(previous_last_state, direction * step_size)
        if counter is not None:
            counter[0] += 1
        proposal, is_div0 = generate_proposal(initial_energy, state)
        momentum_sum = state.momentum
        termination_state = update_termination_state(
            termination_state, momentum_sum, state.momentum, 0
        )
        init = (proposal, state, momentum_sum, termination_state, 1, is_div0, False)
        cur = init
        # scan over steps 1..max_num_steps with `until`: trajectory.py:307-332
        for step in range(1, 1 + max_num_steps):
            prop, last, msum, tstate, length, _, _ = cur
            new_state = integrator(last, direction * step_size)
            if counter is not None and not is_div0:
                counter[0] += 1
            new_prop, is_div = generate_proposal(initial_energy, new_state)
            sampled = progressive_uniform_sampling(g_uniform, prop, new_prop)
            new_msum = msum + new_state.momentum
            new_t = update_termination_state(tstate, new_msum, new_state.momentum, step)
            has_term = is_criterion_met(new_t, new_msum, new_state.momentum)
            cur = (sampled, new_state, new_msum, new_t, length + 1, is_div, has_term)
            if is_div or has_term:
                break
        # trajectory.py:336 -- keep the first-step tuple iff the first step diverged
        return init if is_div0 else cur

    return integrate


# --------------------------------------------------------------------------------------
# a18-a20  multiplicative_expansion + nuts kernel  (trajectory.py:428-714, nuts.py:56-153)
# --------------------------------------------------------------------------------------
def multiplicative_expansion(g_direction, g_biased, trajectory_integrator, uturn_check_fn,
                             max_num_expansions):
    def expand(proposal, left_state, right_state, momentum_sum, termination_state,
               initial_energy, step_size, trace=None):
        out = None
        for step in range(max_num_expansions):
            do_go_right = bernoulli(g_direction, 0.5)  # trajectory.py:516
            direction = 1.0 if do_go_right else -1.0
            start_state = right_state if do_go_right else left_state
            (new_proposal, new_state, subtree_momentum_sum, new_termination_state,
             sub_len, is_div, sub_term) = trajectory_integrator(
                start_state, direction, termination_state, 2 ** step, step_size,
                initial_energy)
            new_left = left_state if do_go_right else new_state
            new_right = new_state if do_go_right else right_state
            new_momentum_sum = momentum_sum + subtree_momentum_sum
            with np.errstate(over="ignore"):
                acceptance_probability = float(np.exp(new_proposal.sum_log_p_accept) / sub_len)
            updated_proposal = proposal._replace(
                sum_log_p_accept=float(np.logaddexp(new_proposal.sum_log_p_accept,
                                                    proposal.sum_log_p_accept)))
            biased = progressive_biased_sampling(g_biased, proposal, new_proposal)  # always drawn
            sampled = updated_proposal if (is_div or sub_term) else biased
            is_turning = uturn_check_fn(new_left.momentum, new_right.momentum, new_momentum_sum)
            proposal, left_state, right_state = sampled, new_left, new_right
            momentum_sum, termination_state = new_momentum_sum, new_termination_state
            out = (proposal, acceptance_probability, step + 1, is_div, is_turning)
            if trace is not None:
                trace.append(dict(direction=direction, sub_len=sub_len,
                                  position=np.array(proposal.state.position), is_div=is_div,
                                  sub_term=sub_term, is_turning=is_turning))
            if is_div or is_turning or sub_term:
                break
        return out

    return expand


def nuts_kernel(srng: RandomStream, target, max_num_expansions=10, divergence_threshold=1000):
    g_momentum = srng.site()   # site #1 nuts.py:113 -> metrics.py:66
    g_direction = srng.site()  # site #2 trajectory.py:516
    g_uniform = srng.site()    # site #3 proposals.py:99
    g_biased = srng.site()     # site #4 proposals.py:131

    def step(state, step_size, inverse_mass_matrix, trace=None):
        momentum_generator, kinetic_energy, uturn_check_fn, velocity = gaussian_metric(
            inverse_mass_matrix)
        integrator = velocity_verlet(target, velocity)
        new_term, update_term, is_crit = iterative_uturn(uturn_check_fn)
        counter = [0]
        integrate = dynamic_integration(g_uniform, integrator, kinetic_energy, update_term,
                                        is_crit, divergence_threshold, counter)
        expand = multiplicative_expansion(g_direction, g_biased, integrate, uturn_check_fn,
                                          max_num_expansions)
        initial_state = state._replace(momentum=momentum_generator(g_momentum))
        initial_termination_state = new_term(initial_state.position, max_num_expansions)
        initial_energy = initial_state.potential_energy + kinetic_energy(initial_state.momentum)
        initial_proposal = ProposalState(initial_state, initial_energy, 0.0, -np.inf)
        proposal, acc, nd, is_div, is_turn = expand(
            initial_proposal, initial_state, initial_state, initial_state.momentum,
            initial_termination_state, initial_energy, step_size, trace)
        return Diagnostics(proposal.state, acc, nd, is_turn, is_div, counter[0])

    return step
