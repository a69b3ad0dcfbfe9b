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

}  // namespace aehmc
