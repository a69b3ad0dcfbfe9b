// Fused, register-resident HMC transition (gfx950): one chain per wavefront, the whole
// hmc.new_kernel(...)(state, eps, imm, L) call in ONE launch.
//
// Covers diagonal / scalar metrics with coordinate-wise targets and D <= 1024 (config
// "100-dim isotropic Gaussian, HMC with 32 leapfrog steps, 4096 chains").  Lane l keeps
// elements l, l+64, ... of q, p, dU/dq in VGPRs for all L steps, so HBM sees only the
// transition's inputs and outputs (q, g in; q, g, p out) instead of 48*D bytes per
// leapfrog; the two energy dot products are __shfl_xor wavefront reductions.
//
// Same arithmetic, in the same order, as the generic lock-step path of engine.cuh
// (tests check the two bit for bit).  Reference: hmc.py:77-124,157-204,
// trajectory.py:31-107, integrators.py:54-73, metrics.py:44-73.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"

namespace aehmc {

struct HmcFusedArgs {
  long long C, D, L;
  double eps, thr;
  const double *eps_c;  // optional per-chain step sizes
  int met_ndim;
  long long imm_cs;     // chain stride of imm / sqrt_mass (0 = shared)
  const double *imm, *sqrt_mass;
  int tkind;
  const double *mu, *sigma, *log_sigma;
  uint64_t *rng;  // [C,2,4]
  double *q, *U, *g;
  aehmc_diagnostics out;
  // multi-transition driver (the user-level scan of tests/test_hmc.py:138-148): T transitions
  // per launch, optional per-transition outputs
  long long T;
  double *samples;   // [T,C,D] or null
  double *acc_hist;  // [T,C] or null
  int32_t *div_hist; // [T,C] or null
};

inline bool target_is_elem_host(int k) {
  return k == AEHMC_T_STD_NORMAL || k == AEHMC_T_ISO_GAUSSIAN || k == AEHMC_T_DIAG_GAUSSIAN;
}
inline bool hmc_fused_supported(int tkind, int met_ndim, long long D) {
  return target_is_elem_host(tkind) && met_ndim < 2 && D <= 1024;
}

template <int R, int TK>
__global__ __launch_bounds__(256) void k_hmc_fused(HmcFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double zlds[];  // [4 waves][R*64]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c = (long long)blockIdx.x * 4 + w;
  if (c >= a.C) return;
  const size_t row = (size_t)c * a.D;
  double *zrow = zlds + (size_t)w * (R * 64);

  double q[R], p[R], g[R], im[R], sm[R], mu[R], sg[R], p0[R], qs[R], gs[R];
  bool ok[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long long i = lane + 64 * r;
    ok[r] = i < a.D;
    const long long ii = ok[r] ? i : 0;
    im[r] = a.imm[c * a.imm_cs + (a.met_ndim == 0 ? 0 : ii)];
    sm[r] = a.sqrt_mass[c * a.imm_cs + (a.met_ndim == 0 ? 0 : ii)];
    mu[r] = TK == AEHMC_T_DIAG_GAUSSIAN ? a.mu[ii] : 0.0;
    sg[r] = TK == AEHMC_T_DIAG_GAUSSIAN ? a.sigma[ii] : 1.0;
    q[r] = ok[r] ? a.q[row + ii] : 0.0;
    g[r] = ok[r] ? a.g[row + ii] : 0.0;
  }
  double U = a.U[c];
  Pcg64 g1 = pcg_load(a.rng + (size_t)c * 8);      // site #1: momentum (hmc.py:122)
  Pcg64 g2 = pcg_load(a.rng + (size_t)c * 8 + 4);  // site #2: accept (hmc.py:194)
  const double eps = a.eps_c ? a.eps_c[c] : a.eps;
  const double b = 0.5 * eps, aa = 1 * eps;
  double pa = 0.0;
  int is_div = 0, acc = 0;

  for (long long t = 0; t < a.T; t++) {
    wave_normals(g1, a.D, [=](long long i, double z) { zrow[i] = z; });  // metrics.py:65-68
    __threadfence_block();
    double kd = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      p[r] = ok[r] ? sm[r] * zrow[lane + 64 * r] : 0.0;
      p0[r] = p[r];
      qs[r] = q[r];
      gs[r] = g[r];
      if (ok[r]) kd += (im[r] * p[r]) * p[r];
    }
    __threadfence_block();
    kd = wave_sum(kd);
    const double H0 = U + 0.5 * kd;  // hmc.py:187

    for (long long l = 0; l < a.L; l++) {  // trajectory.py:86-95, integrators.py:54-73
#pragma unroll
      for (int r = 0; r < R; r++) {
        p[r] = p[r] - b * g[r];
        q[r] = q[r] + aa * (im[r] * p[r]);
        if (TK == AEHMC_T_DIAG_GAUSSIAN) g[r] = ((q[r] - mu[r]) / sg[r]) / sg[r];
        else g[r] = q[r];
        p[r] = p[r] - b * g[r];
      }
    }
    // potential energy at the end point, kinetic energy of the flipped momentum
    double usum = 0.0;
    kd = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (ok[r]) {
        const long long i = lane + 64 * r;
        if (TK == AEHMC_T_STD_NORMAL) usum += 0.5 * (q[r] * q[r]) + AEHMC_LOG_SQRT_2PI;
        else if (TK == AEHMC_T_ISO_GAUSSIAN) usum += q[r] * q[r];
        else {
          double z = (q[r] - mu[r]) / sg[r];
          usum += 0.5 * (z * z) + a.log_sigma[i] + AEHMC_LOG_SQRT_2PI;
        }
        double pf = -1.0 * p[r];  // hmc.py:185
        kd += (im[r] * pf) * pf;
      }
    }
    usum = wave_sum(usum);
    kd = wave_sum(kd);
    const double Unew = a.L > 0 ? (TK == AEHMC_T_ISO_GAUSSIAN ? 0.5 * usum : usum) : U;
    double delta = H0 - (Unew + 0.5 * kd);
    if (isnan(delta)) delta = -INFINITY;
    is_div = fabs(delta) > a.thr;
    pa = exp(delta);
    if (pa > 1.0) pa = 1.0;
    if (pa < 0.0) pa = 0.0;
    acc = rng_bernoulli(g2, pa);  // hmc.py:193-195
    if (acc) {
      U = Unew;
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) {
        q[r] = qs[r];
        g[r] = gs[r];
      }
    }
    if (a.samples) {
      double *dst = a.samples + ((size_t)t * a.C + c) * a.D;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (ok[r]) dst[lane + 64 * r] = q[r];
    }
    if (lane == 0) {
      if (a.acc_hist) a.acc_hist[(size_t)t * a.C + c] = pa;
      if (a.div_hist) a.div_hist[(size_t)t * a.C + c] = is_div;
    }
  }

#pragma unroll
  for (int r = 0; r < R; r++) {
    if (ok[r]) {
      const long long i = lane + 64 * r;
      a.q[row + i] = q[r];
      a.g[row + i] = g[r];
      if (a.out.momentum) a.out.momentum[row + i] = acc ? -1.0 * p[r] : p0[r];
    }
  }
  if (lane == 0) {
    pcg_store(a.rng + (size_t)c * 8, g1);
    pcg_store(a.rng + (size_t)c * 8 + 4, g2);
    a.U[c] = U;
    a.out.acceptance_probability[c] = pa;
    a.out.is_diverging[c] = is_div;
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = a.L * a.T;
    if (a.out.is_turning) a.out.is_turning[c] = acc;  // HMC: reused as the accept flag
  }
}

template <int R>
inline hipError_t launch_hmc_fused_r(const HmcFusedArgs &a, hipStream_t st) {
  dim3 grid((unsigned)((a.C + 3) / 4)), block(256);
  size_t lds = (size_t)4 * R * 64 * sizeof(double);
  switch (a.tkind) {
    case AEHMC_T_STD_NORMAL:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_STD_NORMAL>), grid, block, lds, st, a);
      break;
    case AEHMC_T_ISO_GAUSSIAN:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_ISO_GAUSSIAN>), grid, block, lds, st, a);
      break;
    default:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_DIAG_GAUSSIAN>), grid, block, lds, st, a);
  }
  return hipGetLastError();
}
inline hipError_t launch_hmc_fused(const HmcFusedArgs &a, hipStream_t st) {
  const long long r = (a.D + 63) / 64;
  if (r <= 1) return launch_hmc_fused_r<1>(a, st);
  if (r <= 2) return launch_hmc_fused_r<2>(a, st);
  if (r <= 4) return launch_hmc_fused_r<4>(a, st);
  if (r <= 8) return launch_hmc_fused_r<8>(a, st);
  return launch_hmc_fused_r<16>(a, st);
}

}  // namespace aehmc
