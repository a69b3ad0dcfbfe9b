/* aehmc_hip.h -- C-ABI of the MI355X (gfx950) many-chain HMC/NUTS trajectory engine.
 *
 * Drop-in boundary for the leapfrog hot path of aesara-devs/aehmc.  The
 * reference has no FFI of its own: its boundary is the Python API
 *     aehmc/hmc.py:16   new_state(q, logprob_fn)
 *     aehmc/hmc.py:43   hmc.new_kernel(srng, logprob_fn, divergence_threshold)
 *     aehmc/hmc.py:77   step(state, step_size, inverse_mass_matrix, num_integration_steps)
 *     aehmc/nuts.py:17  nuts.new_kernel(srng, logprob_fn, max_num_expansions, divergence_threshold)
 *     aehmc/nuts.py:56  step(state, step_size, inverse_mass_matrix)
 * and everything those build symbolically (integrators.py, metrics.py, trajectory.py,
 * termination.py, proposals.py).  The entry points below are what a ctypes binding of
 * that path binds; aehmc_amd/{hmc,nuts}.py are thin wrappers.
 *
 * Conventions
 *  - every call returns 0 on success, <0 on error (aehmc_last_error gives the text);
 *    nothing throws across the ABI.  Numerical failure is data (is_diverging), not an
 *    error (reference: proposals.py:43-45, hmc.py:189-191).
 *  - all array arguments are DEVICE pointers owned by the caller (hipMalloc or a
 *    torch-ROCm tensor's data_ptr); chain-major row layout [C, D], float64.  The
 *    library owns only the ctx and the workspace the caller hands it.
 *  - `stream` is a hipStream_t passed as void*; work is enqueued on it in order.  NUTS
 *    polls a pinned "chains still active" word to stop launching early, so a NUTS call
 *    may block the host for part of its duration; results are complete on `stream`.
 *  - a ctx is not thread-safe; distinct ctxs are independent; no global state.  ONE STREAM PER CTX at a time: the
 *    ctx owns scratch that every call rewrites on the call's stream (the packed matrices of the block-resident dense
 *    kernels, per-chain factorisation scratch, the GLM work arrays), so calls on two streams may only share a ctx if
 *    the caller orders them (events); use one ctx per stream otherwise.
 *  - RNG ("scheme A"): per chain, one PCG64 per RNG call site of the
 *    reference graph, in graph-construction order; rng is uint64 [C, n_sites, 4] =
 *    (state_hi, state_lo, inc_hi, inc_lo), advanced in place exactly as numpy's
 *    Generator.normal / Generator.binomial(1, p) would advance it.
 */
#ifndef AEHMC_HIP_H
#define AEHMC_HIP_H

#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime, the math functions and the fixed-width integers itself) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aehmc_ctx aehmc_ctx;

/* logprob_fn stand-ins (arbitrary Python callables cannot be compiled to HIP) */
enum aehmc_target_kind {
  AEHMC_T_STD_NORMAL = 0,   /* aeppl N(0,1) per coordinate: U = sum 0.5 q^2 + log sqrt(2 pi)  (README.md:27-36) */
  AEHMC_T_ISO_GAUSSIAN = 1, /* U = 0.5 |q|^2 (tests/test_trajectory.py:150-151) */
  AEHMC_T_DIAG_GAUSSIAN = 2,/* N(mu, diag sigma^2) */
  AEHMC_T_DENSE_MVN = 3,    /* U = 0.5 (q-mu)^T P (q-mu), P dense symmetric [D,D] */
  AEHMC_T_LINREG = 4,       /* examples/LinearRegression.ipynb:126-166, q = [w, log n] */
  AEHMC_T_CUSTOM = 5,       /* user-defined coordinate-wise target, compiled at run time: aehmc_set_custom_target */
  AEHMC_T_GLM = 6,          /* user-defined row-reduction target over a data matrix: aehmc_set_custom_glm_target */
  AEHMC_T_JOINT = 7         /* user-defined JOINT (non-separable) log-density, D <= 2048, differentiated by the engine:
                               aehmc_set_custom_joint_target */
};

typedef struct {
  int32_t kind;          /* aehmc_target_kind */
  int32_t reserved;
  int64_t D;             /* position dimension */
  const double *mu;      /* [D]   diag / dense */
  const double *sigma;   /* [D]   diag */
  const double *prec;    /* [D,D] dense, row-major */
  const double *X;       /* [N]   linreg (16-byte aligned) */
  const double *y;       /* [N]   linreg (16-byte aligned) */
  int64_t N;
} aehmc_target;

/* gaussian_metric(inverse_mass_matrix) -- metrics.py:10-106.  ndim 0/1/2 = scalar /
 * diagonal / dense exactly as metrics.py:44-63; sqrt_mass is sqrt(1/imm) (ndim<2) or
 * L^-T with imm = L L^T (metrics.py:56-58).  Dense imm must be symmetric. */
typedef struct {
  int32_t ndim;
  int32_t per_chain;       /* 1: imm / sqrt_mass are [C,1] (ndim 0), [C,D] (ndim 1) or [C,D,D]
                              (ndim 2, D <= 2048), one per chain -- what per-chain window
                              adaptation produces; sqrt_mass must be given
                              (aehmc_metric_sqrt_per_chain computes the dense one); 0: shared */
  int64_t D;
  const double *imm;       /* [1] | [D] | [D,D] */
  const double *sqrt_mass; /* [1] | [D] | [D,D], or NULL: aehmc_set_metric computes it on the
                              device (dense: blocked Cholesky + triangular inverse on the fp64
                              MFMA GEMM) into ctx-owned memory */
  int64_t n_chains;        /* per_chain: number of rows C of imm / sqrt_mass (the step calls
                              refuse a different chain count); 0 when shared */
} aehmc_metric;

/* warm-up state of window_adaptation.run (window_adaptation.py:17-116), one row per chain:
 * DualAveragingState (algorithms.py:9-14), Welford state (algorithms.py:141-165) and the
 * current parameters (step_size, inverse_mass_matrix [+ its sqrt-mass]) */
typedef struct {
  int64_t *da_step;                      /* [C] */
  double *da_x, *da_x_avg, *da_g_avg, *da_mu; /* [C] */
  double *wc_mean, *wc_m2;               /* [C,D] (diagonal adaptation) */
  int64_t *wc_n;                         /* [C] */
  double *step_size;                     /* [C] */
  double *imm, *sqrt_mass;               /* [C,D] */
  int32_t full;                          /* 1: is_mass_matrix_full -- wc_m2, imm and sqrt_mass are
                                            [C,D,D] (full covariance per chain, D <= 2048) */
  int32_t reserved;
  double *work;                          /* full && D > 64: [C,D,D] scratch of the window-end
                                            factorisation (smaller D: LDS; may be NULL) */
} aehmc_adapt_state;

/* per-transition outputs == trajectory.py:379-384 Diagnostics (+ n_leapfrog) */
typedef struct {
  double *momentum;               /* [C,D] Diagnostics.state.momentum */
  double *acceptance_probability; /* [C] */
  int64_t *num_doublings;         /* [C] NUTS only (may be NULL for HMC) */
  int32_t *is_turning;            /* [C] NUTS only (may be NULL for HMC) */
  int32_t *is_diverging;          /* [C] */
  int64_t *n_leapfrog;            /* [C] integrator calls that belong to the trajectory */
} aehmc_diagnostics;

int aehmc_create(aehmc_ctx **out, int device);
int aehmc_destroy(aehmc_ctx *ctx);
const char *aehmc_last_error(const aehmc_ctx *ctx);

/* bind logprob_fn / inverse_mass_matrix (device buffers must outlive their use) */
int aehmc_set_target(aehmc_ctx *ctx, const aehmc_target *target);

/* A USER-DEFINED coordinate-wise logprob_fn (hmc.py:16-40 takes any callable; its gradient comes from autodiff,
 * integrators.py:61-65): `source` is HIP source that defines the device function `aehmc_custom_elem`,
 *     __device__ void aehmc_custom_elem
 *         (double q, long long i, const double *const *prm, double &u, double &g)
 * -- the contribution u of coordinate i to the potential energy U = -logprob(q) = sum_i u_i and du_i/dq_i = g --
 * with prm[k] the k-th of `n_params` device arrays (`params`: HOST array of device pointers; [D] each, or whatever the
 * function indexes).  The library compiles its kernel templates against it with hipRTC (libhiprtc, gfx950;
 * `include_dir` = the directory that holds the library's csrc/ headers) on first use and caches the code objects by
 * source: the lock-step engine (any metric, any D), the register-resident NUTS kernel (D <= 512, diagonal / scalar
 * metric) and the fused HMC kernel (D <= 1024, diagonal / scalar metric) and, since round 5, the workgroup-per-chain
 * NUTS / HMC kernels (diagonal / scalar metric, D <= 10176 / 10240) and the block-resident NUTS / HMC kernels (shared
 * dense metric, 64 < D <= 512).  The density alone is enough: with `#include "dual.cuh"` the source may define
 *     template <class T> __device__ T aehmc_logp
 *         (T q, long long i, const double *const *prm)
 * and a three-line aehmc_custom_elem that instantiates it with aehmc::Dual (what aehmc_amd/targets.py appends).
 * Compilation errors come back through aehmc_last_error with the compiler's log, and the previous binding stays. */
int aehmc_set_custom_target(aehmc_ctx *ctx, const char *source, int64_t D, const double *const *params,
                            int32_t n_params, const char *include_dir);

/* A user-defined ROW-REDUCTION ("GLM-type") logprob_fn over a data matrix X [N,D] (row-major, device) and responses
 * y [N] (device):  U(q) = sum_n loss(x_n . q, y_n) + sum_i prior(q_i),  dU/dq = X^T dloss/dz + prior'(q).  `source`
 * defines the device functions `aehmc_glm_row` and `aehmc_glm_prior`,
 *     __device__ void aehmc_glm_row
 *         (double z, double y, long long n, const double *const *prm, double &loss, double &dloss_dz)
 *     __device__ void aehmc_glm_prior
 *         (double q, long long i, const double *const *prm, double &u, double &g)
 * (logistic regression: loss = log1p(exp(z)) - y z, dloss = 1 / (1 + exp(-z)) - y).  Per leapfrog the two products
 * with X run as chain-batched fp64 MFMA GEMMs (Z = Q X^T, then G = dLoss X), the user's functions in run-time
 * compiled kernels between and behind them; lock-step engine (any metric).  The library keeps a transposed copy of X
 * and a [C, N] work array. */
/* (round 5: with D <= 32 and a scalar / diagonal metric NUTS and HMC run whole calls in ONE launch -- the wavefront that owns
 * a chain sweeps the rows itself, k_nuts_glm_rows / k_hmc_glm_rows -- when D <= 16 or the call has <= 1024 chains) */
int aehmc_set_custom_glm_target(aehmc_ctx *ctx, const char *source, int64_t D, int64_t N, const double *X,
                                const double *y, const double *const *params, int32_t n_params,
                                const char *include_dir);

/* A user-defined JOINT logprob_fn (reference: aehmc/hmc.py:16-40 takes any callable and differentiates it,
 * hmc.py:33-34, integrators.py:61-65): the user writes the log-DENSITY only, the engine differentiates it.  `source`
 * is HIP source that includes "dual.cuh" and defines
 *     template <class V> __device__ auto aehmc_logp
 *         (const V &q, const double *const *prm)
 * with q[i] the coordinates (i wave-uniform) and q.size() = D <= 2048 -- hierarchical models, funnels, anything that is
 * not a sum over coordinates or data rows.  Forward mode: lane i of the chain's wavefront evaluates the density with
 * the derivative seeded at coordinate i (csrc/dual.cuh), so for D <= 64 ONE evaluation per leapfrog yields U = -logp and
 * the whole gradient: the single-launch kernels of small problems (k_nuts_resident / k_hmc_fused_dense compiled against
 * it: scalar, diagonal or dense metric, shared or per chain, any number of transitions per launch) and new_state.
 * Above 64 coordinates, and with options resident_nuts / fused_hmc = 0, the density is evaluated on the lock-step path
 * between the stage kernels (k_target_joint_rows: the chain's row in LDS, ceil(D / 64) evaluations per gradient, lane l
 * seeding coordinate l + 64 k in pass k): any metric, O(D^2 / 64) density terms per leapfrog and chain; with a scalar or
 * diagonal metric the same loop runs for one chain per wavefront in one launch per call (k_nuts_joint_rows up to
 * D = 192, k_hmc_joint_rows at any D; bitwise the lock-step path). */
int aehmc_set_custom_joint_target(aehmc_ctx *ctx, const char *source, int64_t D, const double *const *params,
                                  int32_t n_params, const char *include_dir);

/* Code objects of run-time compiled programs (the three entry points above) are kept in `dir` across processes: a file
 * per program, named by a hash of everything the compiler saw (source, options, kernel names).  The caller chooses a
 * directory that is specific to the library's own sources -- the headers a program includes (aehmc_amd/engine.py uses
 * ~/.cache/aehmc_amd/rtc-<source hash>).  NULL or "" switches the cache off (default). */
int aehmc_set_rtc_cache(aehmc_ctx *ctx, const char *dir);
/* How many run-time programs this ctx compiled with hipRTC and how many it took from the directory above (either
 * pointer may be NULL): what a caller checks to know that a second process did not recompile -- a count, not a time. */
int aehmc_rtc_stats(const aehmc_ctx *ctx, int64_t *compiled, int64_t *loaded_from_cache);
int aehmc_set_metric(aehmc_ctx *ctx, const aehmc_metric *metric);

/* engine options (name, default):
 *  "fused_hmc" 1    register-resident single-launch HMC when the metric is diagonal and the
 *                   target coordinate-wise, or the problem small and dense (D <= 64, see
 *                   "resident_nuts"); 0 forces the lock-step path
 *  "resident_nuts" 2 register-resident single-launch NUTS (a team of 1..64 lanes, or a
 *                   256/512-thread workgroup for large D, keeps the chain's moving state on chip
 *                   for the whole tree) for diagonal/scalar metrics, coordinate-wise
 *                   targets, D <= 10176, and the regression target (four chains per workgroup
 *                   share each pass over the data rows), and for small dense problems (D <= 64: a
 *                   dense inverse mass matrix -- shared or one per chain -- and / or the dense-precision
 *                   target, products inside the wavefront).  2 (auto) = 1 = wherever such a kernel exists
 *                   (round 3: it beats the lock-step path at every chain count), 0 = never
 *  "resident_min_team" 0  1: always give a chain the smallest team of lanes that holds it
 *                   (64/T chains per wavefront) instead of widening teams while the GPU would
 *                   otherwise run fewer than ~4096 wavefronts
 *  "fused_nuts" 0   1: whole NUTS transition in one launch (one wavefront per chain loops
 *                   leapfrog + tree bookkeeping over its HBM-resident state) when the metric is
 *                   diagonal and the target coordinate-wise -- lowest latency for a few
 *                   chains; the lock-step path (one launch per leapfrog) has the higher
 *                   throughput for thousands of chains and is the default
 *  "dense_linear" 1 dense metric: carry w = imm g with the state so that
 *                   v_half = v - (eps/2) w, v' = v_half - (eps/2) w' (one metric GEMM per
 *                   leapfrog); 0 forms imm p_half and imm p' directly as metrics.py:71 does
 *  "gemm_small_tiles" 1  fp64 GEMM of a mid-size problem (fewer than 256 tiles of 128 x 128, N <= 2048):
 *                   1 = 64 x 128, 64 x 64 or 32 x 64 tiles, the largest that gives every CU two
 *                   workgroups (bitwise the results of the 128 x 128 kernel); 2 / 3 / 4 force
 *                   64 x 64 / 64 x 128 / 32 x 64; 0 = off
 *  "streamk" 2      fp64 GEMM: persistent grid; whole tiles for all but the last 1..2 rounds, the
 *                   rest of the (tile, k) space split evenly; a tile cut between two workgroups is
 *                   accumulated in k order (bitwise equal to 0).  2: 128 x 256 tiles, one
 *                   workgroup per CU, software-pipelined K loop (default); 1: 128 x 128 tiles,
 *                   two workgroups per CU; 0: one tile per workgroup
 *  "compact" 1      NUTS: chains whose transition has finished drop out of the GEMMs
 *  "block_dense" 1  mid-size dense problems (shared dense inverse mass matrix, 64 < D <= 512, "dense_linear" = 1,
 *                   coordinate-wise or dense-precision target): NUTS ("resident_nuts" != 0) and HMC ("fused_hmc" = 1)
 *                   run the whole call in ONE launch, a workgroup per 16 chains -- stages at a wavefront per chain,
 *                   products on fp64 MFMA inside the workgroup, same k-order as the chain-batched GEMM (bitwise
 *                   the lock-step path's results).  1: the chains' moving state in registers up to D = 256, in
 *                   L2-resident work rows above; 2: work rows at every D; 0 = the lock-step path
 *  "pc_dense" 1     one dense inverse mass matrix PER CHAIN (full-matrix window adaptation), 64 < D <= 512, coordinate-wise
 *                   target, "dense_linear" = 1: NUTS runs the whole call in one launch, the wavefront that owns a chain
 *                   streaming its matrix once per leapfrog (csrc/nuts_pc_dense.cuh; bitwise the lock-step path); 0 =
 *                   the lock-step path (per-chain mat-vec launches between the stage kernels)
 *  "block_roll" 0   block-resident NUTS with the state in registers, launches of several transitions: a chain whose
 *                   tree has ended begins its next transition as soon as this many chains of its workgroup wait
 *                   (one more in-workgroup product in that round: csrc/nuts_block_roll.cuh) instead of waiting for
 *                   the deepest of the 16 trees.  0 = rolling with the kernel's threshold (3 with a dense-precision
 *                   target, 4 otherwise) for D >= 192, all chains of a workgroup transition by transition below;
 *                   1 ... 15 = rolling at every D with this threshold; 16 = never.  Results do not depend on it
 *                   (bitwise)
 *  "fp_contract" 0  1: fast arithmetic in the leapfrog bodies of the register-resident HMC kernels
 *                   (diagonal / scalar metric, coordinate-wise target): every a*b+c one fused multiply-add,
 *                   eps*imm and 1/sigma^2 formed once, the half kicks between consecutive leapfrogs of a
 *                   static trajectory merged -- 2 fp64 operations per element and leapfrog instead of 6.
 *                   The integrator of integrators.py:54-73 within 1e-6 relative (the north star's bar);
 *                   0 (default) rounds every product and sum as the reference does and is bit-identical
 *                   to the oracle.  Momentum draw, energies and accept step are the same code in both modes
 *  "joint_resident" 1  joint user-defined density with a reverse-mode program, D <= 512, scalar / diagonal metric: NUTS on
 *                   the register-resident kernel (the position handed to the program through LDS rows) from 17
 *                   coordinates on or when the density's reductions are long; 2 = at every D <= 512; 0 = never (up to
 *                   64 coordinates the forward-mode dense-path kernel, above the one-launch kernel over the chains' L2
 *                   rows, k_nuts_joint_rows: same arithmetic and bits as that one).  HMC: 64 < D <= 1024 on k_hmc_fused
 *                   compiled against the program in the same way (non-zero), or k_hmc_joint_rows (0): same bits
 *  "joint_wg"    1  joint user-defined density that comes with its reverse-mode program (AEHMC_JOINT_GRAD) and sweeps
 *                   long data (AEHMC_JOINT_SWEEP_TERMS >= 8192) in a call of <= 2048 chains: a WORKGROUP of eight
 *                   wavefronts per chain runs the program (k_nuts_joint_wg / k_hmc_joint_wg); 0 = never (a wavefront
 *                   per chain), 2 = always.  Discrete outputs identical, values to rounding (the sums are associated
 *                   differently).  Also the workgroup-per-chain kernels of a row-reduction target (GLM, N >= 8192)
 *  "wg_waves"    0  wavefronts per SIMD those workgroup-per-chain kernels are compiled for: 4 (two workgroups per CU,
 *                   128 registers per lane), 3 (one, 168 registers); 0 = four unless the program then keeps more than
 *                   320 bytes per lane in scratch.  Same results either way */
int aehmc_set_option(aehmc_ctx *ctx, const char *name, int64_t value);

/* workspace the caller must provide to the step calls for C chains */
int64_t aehmc_workspace_bytes(const aehmc_ctx *ctx, int64_t C, int64_t max_num_expansions);
int aehmc_set_workspace(aehmc_ctx *ctx, void *workspace, int64_t bytes);

/* hmc.new_state -- hmc.py:16-40: U = -logprob(q), g = dU/dq */
int aehmc_new_state(aehmc_ctx *ctx, int64_t C, const double *q, double *U, double *g, void *stream);

/* hmc.new_kernel(...)(state, step_size, imm, L) -- hmc.py:77-124,157-204, trajectory.py:31-107.
 * rng [C,2,4]: site #1 momentum, #2 accept.  q,U,g updated in place. */
int aehmc_hmc_step(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                   int64_t num_integration_steps, double divergence_threshold, double *q,
                   double *U, double *g, const aehmc_diagnostics *out, void *stream);

/* num_samples consecutive HMC transitions per chain in one call -- the user-level loop
 * `aesara.scan(kernel, n_steps=N)` of tests/test_hmc.py:138-148 / README.md.  One launch
 * when the fused path applies.  Optional outputs: samples [N,C,D] (position after every
 * transition), acceptance_history [N,C], divergence_history [N,C]; `out` describes the
 * last transition, out->n_leapfrog the total. */
int aehmc_hmc_sample(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                     int64_t num_integration_steps, double divergence_threshold,
                     int64_t num_samples, double *q, double *U, double *g,
                     const aehmc_diagnostics *out, double *samples, double *acceptance_history,
                     int32_t *divergence_history, void *stream);

/* nuts.new_kernel(...)(state, step_size, imm) -- nuts.py:56-153, trajectory.py:154-374,428-714,
 * termination.py:19-235, proposals.py.  rng [C,4,4]: #1 momentum, #2 direction,
 * #3 uniform progressive, #4 biased progressive.  q,U,g updated in place. */
int aehmc_nuts_step(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                    int64_t max_num_expansions, double divergence_threshold, double *q,
                    double *U, double *g, const aehmc_diagnostics *out, void *stream);

/* per-chain step sizes [n] overriding the scalar step_size argument of the step calls
 * (NULL restores the scalar) -- window adaptation adapts one step size per chain.  A step
 * call with a chain count other than n fails. */
int aehmc_set_step_sizes(aehmc_ctx *ctx, const double *step_sizes, int64_t n);

/* window_adaptation.window_adaptation(...).init / .update (window_adaptation.py:119-227,
 * step_size.py:9-100, mass_matrix.py:12-120, algorithms.py:17-204), diagonal mass matrix,
 * one adaptation per chain.  `stage` / `is_window_end` come from build_schedule
 * (window_adaptation.py:230-327, host side); `is_last` = last warm-up step. */
int aehmc_adapt_init(aehmc_ctx *ctx, int64_t C, int64_t D, double initial_step_size,
                     const aehmc_adapt_state *state, void *stream);
int aehmc_adapt_update(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t stage, int32_t is_window_end,
                       int32_t is_last, double target_acceptance_rate,
                       const double *acceptance_probability, const double *position,
                       const aehmc_adapt_state *state, void *stream);

/* step_size.dual_averaging_adaptation(target, gamma, t0, kappa) -> update (step_size.py:9-100 over
 * algorithms.dual_averaging, algorithms.py:17-115) as a stand-alone building block around any kernel
 * (tests/test_step_size.py:13-88 wraps hmc.new_kernel with it): one update of the C per-chain states
 * (step [C] int64, iterates x = log step size, iterates_avg, gradient_avg, shrinkage_pts mu, all [C],
 * in place) with gradient = target_acceptance_rate - acceptance_probability.  The states start as
 * algorithms.py:56-76 says: step 1, x = x_avg = gradient_avg = 0, mu as given.  `step_size_out` [C]
 * (may be NULL) receives exp(x) of the updated iterate -- what the test feeds to the next transition. */
int aehmc_dual_averaging_update(aehmc_ctx *ctx, int64_t C, double target_acceptance_rate, double gamma,
                                double t0, double kappa, const double *acceptance_probability,
                                int64_t *step, double *iterates, double *iterates_avg, double *gradient_avg,
                                const double *shrinkage_pts, double *step_size_out, void *stream);

/* algorithms.welford_covariance(compute_covariance) -> update / final (algorithms.py:120-204) and
 * mass_matrix.covariance_adaptation -> final (mass_matrix.py:83-118) as stand-alone building blocks, C independent
 * estimators (tests/test_algorithms.py:60-133, tests/test_mass_matrix.py:11-60 drive them directly): `update` takes
 * one new value [C,D] per estimator (mean [C,D], m2 [C,D] -- or [C,D,D] with `full`, grown by
 * outer(updated_delta, delta) --, sample_size [C], all in place); `final` writes m2 / (sample_size - 1) and, with
 * `shrink`, Stan's regularisation (n / (n + 5)) cov + 1e-3 (5 / (n + 5)) (on the diagonal only when `full`) -- the
 * arithmetic of the warm-up kernels, bit for bit. */
int aehmc_welford_update(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t full, const double *value, double *mean,
                         double *m2, int64_t *sample_size, void *stream);
int aehmc_covariance_final(aehmc_ctx *ctx, int64_t C, int64_t D, int32_t full, int32_t shrink, const double *m2,
                           const int64_t *sample_size, double *out, void *stream);

/* window_adaptation.run (window_adaptation.py:17-116) for a NUTS kernel: num_steps x (one transition
 * with the current per-chain parameters, then aehmc_adapt_update), enqueued in one call.  `stage` /
 * `is_window_end` [num_steps] are HOST arrays from build_schedule.  Before the call the caller binds
 * the per-chain metric to state->imm / state->sqrt_mass (aehmc_set_metric, per_chain = 1) and the step
 * sizes to state->step_size (aehmc_set_step_sizes); the update kernel rewrites them in place.
 * `out` receives the diagnostics of the last warm-up transition.  Diagonal mass matrix with the
 * regression target, or with a coordinate-wise target of D <= 512 on the register-resident kernel,
 * and a full mass matrix per chain (state->full) with D <= 64 and a coordinate-wise or dense-precision
 * target, or with the regression target: the whole warm-up is ONE launch in which the chains adapt and move on at their own pace
 * (same arithmetic, same results as the loop). */
int aehmc_nuts_warmup(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t num_steps, const int32_t *stage,
                      const int32_t *is_window_end, double target_acceptance_rate,
                      int64_t max_num_expansions, double divergence_threshold, double *q, double *U,
                      double *g, const aehmc_diagnostics *out, const aehmc_adapt_state *state, void *stream);

/* The same loop around an HMC kernel (the reference's run() calls `kernel(chain_state, *parameters)`,
 * window_adaptation.py:66, so there an HMC kernel is wrapped in a function that closes over the trajectory length
 * num_integration_steps): num_steps x (one HMC transition with the current per-chain parameters, then
 * aehmc_adapt_update).  Same binding rules, same `stage` / `is_window_end` host arrays as aehmc_nuts_warmup. */
int aehmc_hmc_warmup(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t num_steps, const int32_t *stage,
                     const int32_t *is_window_end, double target_acceptance_rate, int64_t num_integration_steps,
                     double divergence_threshold, double *q, double *U, double *g, const aehmc_diagnostics *out,
                     const aehmc_adapt_state *state, void *stream);

/* num_samples consecutive NUTS transitions per chain (the user-level scan of
 * tests/test_hmc.py:296-324); same optional outputs as aehmc_hmc_sample plus the per-chain
 * leapfrog total [C].  `out` describes the last transition.  (Regression target and register-resident
 * kernel, D <= 512: one launch for all num_samples transitions.) */
int aehmc_nuts_sample(aehmc_ctx *ctx, int64_t C, uint64_t *rng, double step_size,
                      int64_t max_num_expansions, double divergence_threshold, int64_t num_samples,
                      double *q, double *U, double *g, const aehmc_diagnostics *out, double *samples,
                      double *acceptance_history, int32_t *divergence_history,
                      int64_t *n_leapfrog_total, void *stream);

/* sqrt_mass = sqrt(1 / imm) (ndim 0 / 1) or chol(imm)^-T (ndim 2; metrics.py:45,49,56-58: blocked Cholesky +
 * triangular inverse on the fp64 MFMA GEMM) of ONE shared metric into a caller-owned array ([1] | [D] | [D,D]) --
 * what aehmc_set_metric computes into ctx memory when aehmc_metric.sqrt_mass is NULL.  A caller that alternates
 * between several metrics factors each once, keeps the results and passes them in aehmc_metric.sqrt_mass. */
int aehmc_metric_sqrt(aehmc_ctx *ctx, int32_t ndim, int64_t D, const double *imm, double *sqrt_mass, void *stream);

/* sqrt_mass[c] = chol(imm[c])^-T (metrics.py:56-58) for C dense D x D matrices, D <= 2048 (one
 * wavefront per matrix: in LDS up to D = 64, in a temporary device buffer above) */
int aehmc_metric_sqrt_per_chain(aehmc_ctx *ctx, int64_t C, int64_t D, const double *imm, double *sqrt_mass,
                                void *stream);

/* ---- building blocks exported for known-answer tests / callers that want them ---- */

/* integrators.py:54-73 applied nsteps times to C chains (state in place) */
int aehmc_leapfrog(aehmc_ctx *ctx, int64_t C, double step_size, int64_t nsteps, double *q,
                   double *p, double *U, double *g, void *stream);
/* metrics.py:70-73 */
int aehmc_kinetic_energy(aehmc_ctx *ctx, int64_t C, const double *p, double *K, void *stream);
/* metrics.py:75-104 */
int aehmc_is_turning(aehmc_ctx *ctx, int64_t C, const double *p_left, const double *p_right,
                     const double *p_sum, int32_t *out, void *stream);
/* RNG streams as numpy would produce them: n normals / bernoulli(p[i]) from rng [C,4] */
int aehmc_rng_normals(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t n, double *out, void *stream);
int aehmc_rng_bernoulli(aehmc_ctx *ctx, int64_t C, uint64_t *rng, int64_t n, const double *p,
                        int32_t *out, void *stream);
/* fp64 MFMA GEMM used by the dense-metric path: Cmat[M,N] = A[M,K] * B[N,K]^T (row-major) */
int aehmc_gemm_nt(aehmc_ctx *ctx, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                  const double *B, int64_t ldb, double *Cmat, int64_t ldc, void *stream);

/* timing hooks for bench.py: HIP events (on the launch stream) around every launch of the
 * dominant kernel (fp64 GEMM, or the fused HMC kernel) since profile_enable(1), and the
 * algorithmic flops of those GEMM launches (2*rows*N*K with the live row count). */
int aehmc_profile_enable(aehmc_ctx *ctx, int enable);
int aehmc_profile_read(aehmc_ctx *ctx, double *kernel_ms_total, int64_t *kernel_launches,
                       double *gemm_flops_total);

/* wait for `stream` and report device-side failures that are not data (a stream-K GEMM
 * hand-off that timed out because the persistent grid was not co-resident): the step calls
 * are asynchronous, so such a failure in the last launches of a call surfaces here, at the
 * next GEMM launch or in aehmc_profile_read. */
int aehmc_synchronize(aehmc_ctx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AEHMC_HIP_H */
