// User-defined row-reduction ("GLM-type") targets, compiled at run time only (aehmc_set_custom_glm_target):
//     U(q) = sum_n loss(x_n . q, y_n) + sum_i prior(q_i),   dU/dq = X^T dloss/dz + prior'(q)
// with the user's two device functions
//     __device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &loss, double &dloss)
//     __device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g)
// The two products with the data matrix (Z = Q X^T over all chains, G = dLoss X) are chain-batched fp64 MFMA GEMMs of
// the prebuilt library (gemm_f64.cuh); the kernels below are what lies between them and behind them.
// Reference: hmc.py:16-40 (any logprob_fn), integrators.py:61-65 (its gradient).
#pragma once
#include "engine.cuh"

namespace aehmc {

// one wavefront per (live) chain: Z[c, n] <- dloss/dz at z = Z[c, n], lsum[c] = sum_n loss (each lane adds its rows
// in ascending order, then the wave sum)
__global__ __launch_bounds__(256) void k_glm_rows(long long C, long long N, const double *y, const double *const *prm,
                                                  double *Z, double *lsum, const int *row_idx, const int *n_rows) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long long c = w;
  if (row_idx) {
    if (w >= *n_rows) return;
    c = row_idx[w];
  } else if (w >= C) {
    return;
  }
  double *z = Z + (size_t)c * N;
  double s = 0.0;
  for (long long n = lane; n < N; n += 64) {
    double l, d;
    aehmc_glm_row(z[n], y[n], n, prm, l, d);
    z[n] = d;
    s += l;
  }
  s = wave_sum(s);
  if (lane == 0) lsum[c] = s;
}
// g[c, i] += prior'(q_i); U = lsum[c] + sum_i prior(q_i) -> ctl[c].U_cur (leapfrog) or U[c] (new_state)
__global__ __launch_bounds__(256) void k_glm_finish(EngineArgs a, const double *q, double *g, double *U, const double *lsum,
                                                    int to_ctl) {
  AEHMC_CHAIN_OF_WAVE();
  if (to_ctl && a.ctl[c].done) return;
  const size_t row = (size_t)c * a.D;
  double us = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double u, pg;
    aehmc_glm_prior(q[row + i], i, a.cparams, u, pg);
    g[row + i] = g[row + i] + pg;
    us += u;
  }
  us = wave_sum(us);
  if (lane == 0) {
    const double Uv = lsum[c] + us;
    if (to_ctl) a.ctl[c].U_cur = Uv;
    else U[c] = Uv;
  }
}

// ---- small-D row-reduction targets in ONE launch per call (round 5) ---------------------------------------------------
// The lock-step path above pays two chain-batched GEMMs and three more launches per leapfrog -- right for large D x N,
// ~80 us per leapfrog whatever the size.  With D = DA <= 32 coordinates (the kernels are compiled at run time, for the
// target's own D) the wavefront that owns a chain can sweep the data itself: lane l takes rows l, l + 64, ... (X^T [D][N]: coalesced), forms z_n = sum_d x_nd q_d with the
// position in LDS, calls the user's row function and keeps D partial sums of x_nd dloss_n in registers; D wave sums and
// the prior finish U and dU/dq.  Around it the lock-step engine's own stage / bookkeeping device functions, one chain
// per wavefront (as k_nuts_fused / k_nuts_pc_dense / k_nuts_joint_rows).  The sums run in another order than the GEMMs':
// results agree with the lock-step path to rounding (1e-13), with the numpy restatement at the usual 1e-9.
// Wavefronts per SIMD the one-launch kernels leave room for.  Instantiated for the target's own D the row loop has no
// predicates and the compiler, left alone, spends registers on it (k_nuts_glm_rows<16>: 255, ONE wavefront per SIMD where
// 4096 chains are four).  Logistic regression N = 1e4, 4096 chains, NUTS, ms per transition at 1 / 2 / 3 / 4 wavefronts asked
// for: D = 8 8.7 / 8.9 / 8.8 / 8.2, D = 16 23.2 / 18.5 / 18.7 / 22.7 (profiles/r6/INDEX.md).
constexpr int glm_rows_min_waves(int DA) { return DA <= 8 ? 4 : 2; }
template <int DA>
__device__ inline double glm_rows_eval(const EngineArgs &a, const double *XT, const double *y, long long N, const double *q,
                                       double *g, double *qs, int lane) {
  constexpr int D = DA;  // (the engine instantiates the kernel for the target's own D: no predicates in the row loop)
  for (int d = lane; d < D; d += 64) qs[d] = q[d];
  __threadfence_block();  // (read back as broadcasts)
  double acc[DA], ls = 0.0;
#pragma unroll
  for (int d = 0; d < DA; d++) acc[d] = 0.0;
  // UN row groups per trip, all their loads requested before the first use
  // (measured: 2 and 4 groups per trip change nothing -- the sweep is bound by the user's row function, ~200 vector
  //  instructions per 64 rows for a logistic loss in forward mode, not by the round trips -- and cost registers)
  constexpr int UN = 1;
  for (long long n0 = lane; n0 < N; n0 += 64 * UN) {
    double x[UN][DA], yv[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long long n = n0 + 64 * u, nc = n < N ? n : N - 1;  // (past the end: a valid row, its terms dropped below)
#pragma unroll
      for (int d = 0; d < DA; d++) x[u][d] = d < D ? XT[(size_t)d * N + nc] : 0.0;
      yv[u] = y[nc];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long long n = n0 + 64 * u;
      double z = 0.0;
#pragma unroll
      for (int d = 0; d < DA; d++)
        if (d < D) z += x[u][d] * qs[d];
      double l, dl;
      aehmc_glm_row(z, yv[u], n < N ? n : N - 1, a.cparams, l, dl);
      if (n >= N) l = dl = 0.0;
      ls += l;
#pragma unroll
      for (int d = 0; d < DA; d++) acc[d] += x[u][d] * dl;
    }
  }
  ls = wave_sum(ls);
  double gl = 0.0;  // lane d ends with dU/dq_d
#pragma unroll
  for (int d = 0; d < DA; d++) {
    if (d < D) {
      const double s = wave_sum(acc[d]);
      if (lane == d) gl = s;
    }
  }
  double us = 0.0;
  if (lane < D) {
    double u, pg;
    aehmc_glm_prior(qs[lane], lane, a.cparams, u, pg);
    g[lane] = gl + pg;
    us = u;
  }
  us = wave_sum(us);
  __threadfence_block();  // (the stage that follows reads g through other lanes' addresses)
  return ls + us;
}
template <int DA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(glm_rows_min_waves(DA)))) void k_nuts_glm_rows(EngineArgs a, NutsSampleArgs m, const double *XT, const double *y, long long N) {
  __shared__ double glm_q[4][DA];
  AEHMC_CHAIN_OF_WAVE();
  double *const qs = glm_q[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const size_t row = (size_t)c * a.D;
  ChainRng rng = rng_load(a, c);
  ChainCtl ct = {};
  double U_state = a.U[c];
  long long nleap_sum = 0;
  for (long long t_idx = 0; t_idx < m.T; t_idx++) {
    draw_momentum<false>(a, c, lane, rng.g[0]);
    nuts_init_chain<false>(a, c, lane, ct, rng, &U_state);
    while (!ct.done) {
      double U_new = 0.0;
      leap_stages<true, true, false, false>(a, c, lane, ct.dir, U_new);  // p_half, q'
      ct.U_cur = glm_rows_eval<DA>(a, XT, y, N, a.cur_q + row, a.cur_g + row, qs, lane);
      leap_stages<false, false, true, false>(a, c, lane, ct.dir, U_new);  // p' = p_half - b dU/dq'
      nuts_book<false>(a, c, lane, ct, rng);
    }
    U_state = pick2(ct.U_slot, ct.prop_slot);
    nleap_sum += ct.nleap;
    __threadfence_block();
    if (m.samples) {
      double *dst = m.samples + ((size_t)t_idx * a.C + c) * a.D;
      for (long long i = lane; i < a.D; i += 64) dst[i] = a.q[row + i];
    }
    if (lane == 0) {
      if (m.acc_hist) m.acc_hist[(size_t)t_idx * a.C + c] = ct.acc_prob;
      if (m.div_hist) m.div_hist[(size_t)t_idx * a.C + c] = ct.out_div;
    }
  }
  rng_store(a, c, lane, rng, 0, 3);
  if (lane == 0 && m.nleap_total) m.nleap_total[c] = nleap_sum;
}
template <int DA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(glm_rows_min_waves(DA)))) void k_hmc_glm_rows(EngineArgs a, long long L, long long nt, double *samples, double *acc_hist,
                                                      int *div_hist, const double *XT, const double *y, long long N) {
  __shared__ double glm_q[4][DA];
  AEHMC_CHAIN_OF_WAVE();
  double *const qs = glm_q[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const size_t row = (size_t)c * a.D;
  Pcg64 g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4), g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
  double U_state = a.U[c];
  for (long long tt = 0; tt < nt; tt++) {
    draw_momentum<false>(a, c, lane, g1);
    ChainCtl ct = hmc_init_chain<false>(a, c, lane, &U_state);
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95
      double U_new = 0.0;
      leap_stages<true, true, false, false>(a, c, lane, 1, U_new);
      ct.U_cur = glm_rows_eval<DA>(a, XT, y, N, a.cur_q + row, a.cur_g + row, qs, lane);
      leap_stages<false, false, true, false>(a, c, lane, 1, U_new);
    }
    __threadfence_block();
    const HmcEnd e = hmc_end_chain_rng<false>(a, c, lane, ct, L, g2);
    if (e.acc) U_state = ct.U_cur;
    if (samples) {
      double *dst = samples + ((size_t)tt * a.C + c) * a.D;
      for (long long i = lane; i < a.D; i += 64) dst[i] = a.q[row + i];
    }
    if (lane == 0) {
      if (acc_hist) acc_hist[(size_t)tt * a.C + c] = e.pa;
      if (div_hist) div_hist[(size_t)tt * a.C + c] = e.is_div;
    }
  }
  if (lane == 0) {
    pcg_store(a.rng + (size_t)c * a.nsites * 4, g1);
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, g2);
  }
}

// ---- a WORKGROUP per chain (round 6): long data, few chains.  With <= 2048 chains a wavefront per chain leaves the GPU's
// 1024 SIMDs with one or two wavefronts each, every one walking all N rows (latency-bound: profiles/r6/INDEX.md).  Here W
// wavefronts share a chain's sweep -- thread t takes rows t, t + 64 W, ... --, their partial sums meet in LDS (added in
// wavefront order by every thread: the same bits everywhere), and wavefront 0 runs the chain's stage / bookkeeping
// functions between the sweeps.  The sums are associated differently than in glm_rows_eval: results agree to rounding.
template <int DA, int W>
__device__ inline double glm_rows_eval_wg(const EngineArgs &a, const double *XT, const double *y, long long N, const double *q,
                                          double *g, double *qs, double *part, int tid) {
  constexpr int D = DA;
  const int lane = tid & 63, wave = tid >> 6;
  if (tid < D) qs[tid] = q[tid];
  __syncthreads();
  double acc[DA], ls = 0.0;
#pragma unroll
  for (int d = 0; d < DA; d++) acc[d] = 0.0;
  for (long long n = tid; n < N; n += 64 * W) {
    double x[DA];
#pragma unroll
    for (int d = 0; d < DA; d++) x[d] = d < D ? XT[(size_t)d * N + n] : 0.0;
    double z = 0.0;
#pragma unroll
    for (int d = 0; d < DA; d++)
      if (d < D) z += x[d] * qs[d];
    double l, dl;
    aehmc_glm_row(z, y[n], n, a.cparams, l, dl);
    ls += l;
#pragma unroll
    for (int d = 0; d < DA; d++) acc[d] += x[d] * dl;
  }
  ls = wave_sum(ls);
#pragma unroll
  for (int d = 0; d < DA; d++)
    if (d < D) acc[d] = wave_sum(acc[d]);
  if (lane == 0) {
    part[wave * (DA + 1)] = ls;
#pragma unroll
    for (int d = 0; d < DA; d++) part[wave * (DA + 1) + 1 + d] = acc[d];
  }
  __syncthreads();
  double tot = part[0];
  for (int w = 1; w < W; w++) tot += part[w * (DA + 1)];
  double us = 0.0;
  if (tid < D) {  // thread d finishes dU/dq_d (wavefront 0: D <= 32)
    double gl = part[1 + tid];
    for (int w = 1; w < W; w++) gl += part[w * (DA + 1) + 1 + tid];
    double u, pg;
    aehmc_glm_prior(qs[tid], tid, a.cparams, u, pg);
    g[tid] = gl + pg;
    us = u;
  }
  if (wave == 0) {
    us = wave_sum(us);
    if (lane == 0) part[W * (DA + 1)] = us;
  }
  __threadfence_block();  // (wavefront 0's stage functions read g behind the caller's barrier)
  __syncthreads();
  return tot + part[W * (DA + 1)];
}
template <int DA, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(wg_min_waves(W)))) void k_nuts_glm_wg(EngineArgs a, NutsSampleArgs m, const double *XT, const double *y, long long N) {
  __shared__ double glm_q[DA];
  __shared__ double glm_part[W * (DA + 1) + 1];
  __shared__ int wg_done;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool leader = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const long long c = blockIdx.x;
  const size_t row = (size_t)c * a.D;
  ChainRng rng = rng_load(a, c);
  ChainCtl ct = {};
  double U_state = a.U[c];
  long long nleap_sum = 0;
  for (long long t_idx = 0; t_idx < m.T; t_idx++) {
    if (leader) {
      draw_momentum<false>(a, c, lane, rng.g[0]);
      nuts_init_chain<false>(a, c, lane, ct, rng, &U_state);
    }
    for (;;) {
      double U_new = 0.0;
      if (leader) {
        leap_stages<true, true, false, false>(a, c, lane, ct.dir, U_new);  // p_half, q'
        __threadfence_block();
      }
      __syncthreads();
      const double Uv = glm_rows_eval_wg<DA, W>(a, XT, y, N, a.cur_q + row, a.cur_g + row, glm_q, glm_part, tid);
      if (leader) {
        ct.U_cur = Uv;
        leap_stages<false, false, true, false>(a, c, lane, ct.dir, U_new);  // p' = p_half - b dU/dq'
        nuts_book<false>(a, c, lane, ct, rng);
        if (lane == 0) wg_done = ct.done;
      }
      __syncthreads();
      if (wg_done) break;
    }
    if (leader) {
      U_state = pick2(ct.U_slot, ct.prop_slot);
      nleap_sum += ct.nleap;
      __threadfence_block();
      if (m.samples) {
        double *dst = m.samples + ((size_t)t_idx * a.C + c) * a.D;
        for (long long i = lane; i < a.D; i += 64) dst[i] = a.q[row + i];
      }
      if (lane == 0) {
        if (m.acc_hist) m.acc_hist[(size_t)t_idx * a.C + c] = ct.acc_prob;
        if (m.div_hist) m.div_hist[(size_t)t_idx * a.C + c] = ct.out_div;
      }
    }
  }
  if (leader) {
    rng_store(a, c, lane, rng, 0, 3);
    if (lane == 0 && m.nleap_total) m.nleap_total[c] = nleap_sum;
  }
}
template <int DA, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(wg_min_waves(W)))) void k_hmc_glm_wg(EngineArgs a, long long L, long long nt, double *samples, double *acc_hist,
                                                       int *div_hist, const double *XT, const double *y, long long N) {
  __shared__ double glm_q[DA];
  __shared__ double glm_part[W * (DA + 1) + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const bool leader = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const long long c = blockIdx.x;
  const size_t row = (size_t)c * a.D;
  Pcg64 g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4), g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
  double U_state = a.U[c];
  for (long long tt = 0; tt < nt; tt++) {
    ChainCtl ct = {};
    if (leader) {
      draw_momentum<false>(a, c, lane, g1);
      ct = hmc_init_chain<false>(a, c, lane, &U_state);
    }
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95
      double U_new = 0.0;
      if (leader) {
        leap_stages<true, true, false, false>(a, c, lane, 1, U_new);
        __threadfence_block();
      }
      __syncthreads();
      const double Uv = glm_rows_eval_wg<DA, W>(a, XT, y, N, a.cur_q + row, a.cur_g + row, glm_q, glm_part, tid);
      if (leader) {
        ct.U_cur = Uv;
        leap_stages<false, false, true, false>(a, c, lane, 1, U_new);
      }
    }
    if (leader) {
      __threadfence_block();
      const HmcEnd e = hmc_end_chain_rng<false>(a, c, lane, ct, L, g2);
      if (e.acc) U_state = ct.U_cur;
      if (samples) {
        double *dst = samples + ((size_t)tt * a.C + c) * a.D;
        for (long long i = lane; i < a.D; i += 64) dst[i] = a.q[row + i];
      }
      if (lane == 0) {
        if (acc_hist) acc_hist[(size_t)tt * a.C + c] = e.pa;
        if (div_hist) div_hist[(size_t)tt * a.C + c] = e.is_div;
      }
    }
  }
  if (leader && lane == 0) {
    pcg_store(a.rng + (size_t)c * a.nsites * 4, g1);
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, g2);
  }
}

}  // namespace aehmc
