// NUTS / HMC with ONE DENSE INVERSE MASS MATRIX PER CHAIN, 64 < D <= 512, in one launch (gfx950).
//
// What `window_adaptation.run(is_mass_matrix_full=True)` hands back is a [C, D, D] array of metrics
// (/root/reference/aehmc/mass_matrix.py:12-120, window_adaptation.py:119-227: adaptation is per chain).  No GEMM exists
// here -- every chain has its own matrix -- so a leapfrog is bound by streaming that matrix from HBM: in linear dense
// mode (engine.cuh leap_linear: v = imm p carried by linearity) ONE product w' = imm dU/dq' per leapfrog, D^2 x 8 bytes
// per chain.  Up to D = 64 k_nuts_resident's DENSE instantiations keep the matrix in LDS; above, the lock-step engine
// ran k_matvec_pc_rows launches between its stage kernels, every leapfrog (four launches, the chains of the deepest
// tree holding everybody).  Here the wavefront that owns a chain runs the whole call -- any number of transitions --
// with the lock-step engine's own device functions (leap_linear, nuts_book, nuts_init_chain: the chain's vectors are
// rows of the same L2-resident work arrays) and forms its products itself: a matrix row is read by the 64 lanes in
// whole cache lines, eight rows in flight per wavefront (4096 wavefronts x 16 KB: far beyond the bandwidth-delay
// product of HBM), each lane adding its elements in ascending order and the row finished by wave_sum -- the order of
// k_matvec_pc_rows, hence BITWISE the lock-step path's results (tests/test_gpu_pc_dense.py).
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235, integrators.py:54-73,
// metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"

namespace aehmc {

constexpr int PCD_MIN_D = 65, PCD_MAX_D = 512;
inline bool nuts_pc_dense_supported(int tkind, int met_ndim, int per_chain, long long D) {
  const bool elem = tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN;
  return met_ndim == 2 && per_chain && elem && D >= PCD_MIN_D && D <= PCD_MAX_D;
}

// y[i] = sum_j M[i][j] x[j], i < D, for one wavefront: M row-major [D][D] in global memory, x and y rows of work
// arrays (element lane + 64 r in slot r).  R: slots per lane (D <= 64 R).
template <int R>
__device__ __forceinline__ void wave_matvec_stream(const double *__restrict__ M, const double *x, double *y, int D, int lane) {
  double xr[R], yr[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    xr[r] = (lane + 64 * r < D) ? x[lane + 64 * r] : 0.0;
    yr[r] = 0.0;
  }
  constexpr int NR = 8;  // rows in flight
  // (columns past D: the clamped address re-reads an element of the row, the operand there is 0)
  int off[R];
#pragma unroll
  for (int r = 0; r < R; r++) off[r] = (lane + 64 * r < D) ? lane + 64 * r : lane % D;
  for (int i0 = 0; i0 < D; i0 += NR) {
    double s[NR];
#pragma unroll
    for (int u = 0; u < NR; u++) {
      const int i = i0 + u < D ? i0 + u : D - 1;  // (past the last row: computed again, not used)
      const double *row = M + (size_t)i * D;
      double mv[R];
#pragma unroll
      for (int r = 0; r < R; r++) mv[r] = row[off[r]];
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) acc += mv[r] * xr[r];
      s[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < NR; u++) {
      const double t = wave_sum(s[u]);
      const int i = i0 + u;
      if (i < D) {
#pragma unroll
        for (int r = 0; r < R; r++)
          if (lane + 64 * r == i) yr[r] = t;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if (lane + 64 * r < D) y[lane + 64 * r] = yr[r];
  __threadfence_block();  // (the stage functions read y through other lanes' addresses)
}

// nuts_run's lock-step loop (engine.hip) for ONE chain per wavefront, in one launch: same device functions, same
// products (linear dense mode, k_matvec_pc_rows' summation order), hence the same bits.  m.T transitions per launch.
template <int R>
__global__ __launch_bounds__(256) void k_nuts_pc_dense(EngineArgs a, NutsSampleArgs m) {
  AEHMC_CHAIN_OF_WAVE();
  const int D = (int)a.D;
  const size_t row = (size_t)c * a.D;
  const double *const imm = a.imm + (size_t)c * D * D, *const sm = a.sqrt_mass + (size_t)c * D * D;
  ChainRng rng = rng_load(a, c);
  ChainCtl ct = {};
  double U_state = a.U[c];
  long long nleap_sum = 0;
  for (long long t_idx = 0; t_idx < m.T; t_idx++) {
    // ---- momentum: p = L^-T z (metrics.py:65-68), v = imm p, w = imm dU/dq (nuts.py:113-125) ----
    draw_momentum<true>(a, c, lane, rng.g[0]);
    wave_matvec_stream<R>(sm, a.zbuf + row, a.cur_p + row, D, lane);
    wave_matvec_stream<R>(imm, a.cur_p + row, a.cur_v + row, D, lane);
    wave_matvec_stream<R>(imm, a.g + row, a.cur_w + row, D, lane);
    nuts_init_chain<true>(a, c, lane, ct, rng, &U_state);
    {
      double U_next = 0.0;
      if (leap_linear<12>(a, c, lane, ct.dir, U_next)) ct.U_cur = U_next;
    }
    // ---- one leapfrog per trip: w' = imm dU/dq' | last stage + bookkeeping + the next leapfrog's first stages ----
    while (!ct.done) {
      __threadfence_block();
      wave_matvec_stream<R>(imm, a.cur_g + row, a.cur_w + row, D, lane);
      nuts_book<true, 1>(a, c, lane, ct, rng);
      if (!ct.done) {
        __threadfence_block();
        double U_next = 0.0;
        if (leap_linear<12>(a, c, lane, ct.dir, U_next)) ct.U_cur = U_next;
      }
    }
    // ---- per-transition records (the outputs themselves were written by nuts_write_outputs) ----
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

// HMC: hmc_run's lock-step loop (engine.hip) for one chain per wavefront, nt transitions x L leapfrogs in one launch.
template <int R>
__global__ __launch_bounds__(256) void k_hmc_pc_dense(EngineArgs a, long long L, long long nt, double *samples,
                                                       double *acc_hist, int *div_hist) {
  AEHMC_CHAIN_OF_WAVE();
  const int D = (int)a.D;
  const size_t row = (size_t)c * a.D;
  const double *const imm = a.imm + (size_t)c * D * D, *const sm = a.sqrt_mass + (size_t)c * D * D;
  Pcg64 g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4), g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
  double U_state = a.U[c];
  for (long long tt = 0; tt < nt; tt++) {
    draw_momentum<true>(a, c, lane, g1);
    wave_matvec_stream<R>(sm, a.zbuf + row, a.cur_p + row, D, lane);   // p = L^-T z
    wave_matvec_stream<R>(imm, a.cur_p + row, a.cur_v + row, D, lane);  // v = imm p
    wave_matvec_stream<R>(imm, a.g + row, a.cur_w + row, D, lane);      // w = imm dU/dq
    ChainCtl ct = hmc_init_chain<true>(a, c, lane, &U_state);
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95
      double U_new = 0.0;
      if (leap_linear<12>(a, c, lane, 1, U_new)) ct.U_cur = U_new;
      __threadfence_block();
      wave_matvec_stream<R>(imm, a.cur_g + row, a.cur_w + row, D, lane);  // w' = imm dU/dq'
      if (leap_linear<3>(a, c, lane, 1, U_new)) ct.U_cur = U_new;
    }
    __threadfence_block();
    const HmcEnd e = hmc_end_chain_rng<true>(a, c, lane, ct, L, g2);
    if (e.acc) U_state = ct.U_cur;
    if (samples) {  // (a.q: what this lane has just written, or left untouched on rejection)
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

inline hipError_t launch_nuts_pc_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) {
  const dim3 grid((unsigned)((a.C + 3) / 4)), block(256);
  const int R = (int)((a.D + 63) / 64);
  if (R <= 2) hipLaunchKernelGGL((k_nuts_pc_dense<2>), grid, block, 0, st, a, m);
  else if (R <= 4) hipLaunchKernelGGL((k_nuts_pc_dense<4>), grid, block, 0, st, a, m);
  else if (R <= 6) hipLaunchKernelGGL((k_nuts_pc_dense<6>), grid, block, 0, st, a, m);
  else hipLaunchKernelGGL((k_nuts_pc_dense<8>), grid, block, 0, st, a, m);
  return hipGetLastError();
}

inline hipError_t launch_hmc_pc_dense(const EngineArgs &a, long long L, long long nt, double *samples, double *acc_hist,
                                      int *div_hist, hipStream_t st) {
  const dim3 grid((unsigned)((a.C + 3) / 4)), block(256);
  const int R = (int)((a.D + 63) / 64);
  if (R <= 2) hipLaunchKernelGGL((k_hmc_pc_dense<2>), grid, block, 0, st, a, L, nt, samples, acc_hist, div_hist);
  else if (R <= 4) hipLaunchKernelGGL((k_hmc_pc_dense<4>), grid, block, 0, st, a, L, nt, samples, acc_hist, div_hist);
  else if (R <= 6) hipLaunchKernelGGL((k_hmc_pc_dense<6>), grid, block, 0, st, a, L, nt, samples, acc_hist, div_hist);
  else hipLaunchKernelGGL((k_hmc_pc_dense<8>), grid, block, 0, st, a, L, nt, samples, acc_hist, div_hist);
  return hipGetLastError();
}

}  // namespace aehmc
