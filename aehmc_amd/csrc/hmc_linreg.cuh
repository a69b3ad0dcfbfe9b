// Fused HMC for the regression target (examples/LinearRegression.ipynb:126-166, q = [w, log n],
// D = 2): T transitions x (momentum draw, L leapfrogs, accept) of four chains per 512-thread
// workgroup in ONE launch -- the notebook's own setting is L = 1024 (:43-47), which the
// lock-step path runs as 4 x 1024 dependent launches per transition.
//
// The gradient of this target is a reduction over the N data rows, so the workgroup evaluates
// it cooperatively (linreg_rows.cuh): waves 0..3 each own a chain (lanes 0 and 1 hold its two
// coordinates), all eight waves add their share of the rows for the four chains.  When the rows
// fit (N <= 10176: the notebook's 1e4 rows) X and y are copied into LDS once and stay there for
// every leapfrog of every transition; otherwise they stream from L2 through the LDS-DMA ring of
// the NUTS kernel (config c5's 1e5 rows).  Two barriers per leapfrog.
//
// Diagonal / scalar metric, or (round 3) a dense 2 x 2 inverse mass matrix, shared or one per chain: row `lane` of
// the matrix and of L^-T in two registers, velocities formed literally (metrics.py:71), as in k_nuts_linreg<DM>.
// Same arithmetic as the lock-step path except for the order of the row sums (1e-13).
// Reference: hmc.py:77-124,157-204, trajectory.py:31-107, integrators.py:54-73, metrics.py:44-73.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"
#include "hmc_fused.cuh"
#include "linreg_rows.cuh"

namespace aehmc {

constexpr long long HMC_LINREG_LDS_ROWS = 10176;  // 2 x 8 B per row next to ~1 KB of static LDS

// K: chains per workgroup the arithmetic is laid out for (4; 1..3 only for a call with fewer than
// four chains -- the ragged last workgroup of a larger call carries zero-weight ghosts instead)
template <bool RES, int K>
__global__ __launch_bounds__(LR_BLOCK) void k_hmc_linreg(HmcFusedArgs a) {
  __shared__ double lr_w[4], lr_part[LR_WAVES][8];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long c0 = (long long)blockIdx.x * 4;
  const long long c = wave < 4 ? c0 + wave : a.C;
  const bool ghost = c >= a.C;  // a wave without a chain only serves rows
  const bool ok = !ghost && lane < 2;
  const double N = (double)a.N;
  if (RES) {  // the data rows, once
    for (long long i = tid; i < a.N; i += LR_BLOCK) {
      dyn_lds[i] = a.X[i];
      dyn_lds[a.N + i] = a.y[i];
    }
    __syncthreads();
  }
  double q = 0.0, g = 0.0, im = 1.0, sm = 1.0, U = 0.0, eps = 0.0;
  double im0 = 0.0, im1 = 0.0, sm0 = 0.0, sm1 = 0.0;  // dense metric: row `lane` of the matrix and of L^-T
  const bool dm = a.met_ndim == 2;                     // (a kernel argument: wave-uniform)
  // velocity element `lane` of the momentum whose elements sit in lanes 0, 1 of pv
  auto vel = [&](double pv) -> double {
    if (!dm) return im * pv;
    const double pv0 = read_lane_f64(pv, 0), pv1 = read_lane_f64(pv, 1);
    return im0 * pv0 + im1 * pv1;
  };
  Pcg64 g1{}, g2{};
  if (!ghost) {
    if (dm) {  // [2, 2] shared, or [C, 2, 2]
      const size_t mo = (a.imm_cs ? (size_t)c * 4 : 0) + 2 * (lane < 2 ? lane : 0);
      im0 = a.imm[mo];
      im1 = a.imm[mo + 1];
      sm0 = a.sqrt_mass[mo];
      sm1 = a.sqrt_mass[mo + 1];
    } else {
      const size_t mo = (size_t)c * a.imm_cs + (a.met_ndim == 0 ? 0 : (lane < 2 ? lane : 0));
      im = a.imm[mo];
      sm = a.sqrt_mass[mo];
    }
    q = ok ? a.q[c * 2 + lane] : 0.0;
    g = ok ? a.g[c * 2 + lane] : 0.0;
    U = a.U[c];
    g1 = pcg_load(a.rng + (size_t)c * 8);      // site #1: momentum (hmc.py:122)
    g2 = pcg_load(a.rng + (size_t)c * 8 + 4);  // site #2: accept (hmc.py:194)
    eps = a.eps_c ? a.eps_c[c] : a.eps;
  }
  const double b = 0.5 * eps, aa = 1 * eps;
  double p = 0.0, p0 = 0.0, pa = 0.0;
  int is_div = 0, acc = 0;

  for (long long t = 0; t < a.T; t++) {
    double kd = 0.0, Unew = U;
    const double qs = q, gs = g;
    if (!ghost) {  // metrics.py:65-68: z ~ normal(size=2), two consecutive draws of site #1
      const double z0 = rng_standard_normal(g1), z1 = rng_standard_normal(g1);
      p = ok ? (dm ? sm0 * z0 + sm1 * z1 : sm * (lane == 0 ? z0 : z1)) : 0.0;
      p0 = p;
      const double v_init = vel(p);
      kd = wave_sum(ok ? v_init * p : 0.0);
    }
    const double H0 = U + 0.5 * kd;  // hmc.py:187
    for (long long l = 0; l < a.L; l++) {  // trajectory.py:86-95, integrators.py:54-73
      if (dm) {
        const double ph = p - b * g;
        const double vh = vel(ph);  // (both elements of p_half: outside the lane mask)
        if (ok) {
          p = ph;
          q = q + aa * vh;
        }
      } else if (ok) {
        p = p - b * g;
        q = q + aa * (im * p);
      }
      if (lane == 0 && wave < 4) lr_w[wave] = q;
      __syncthreads();
      double w4[4], sxr[4], srr[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        w4[k] = lr_w[k];
        sxr[k] = srr[k] = 0.0;
      }
      // K == 1: the chain's wave evaluates the row-independent part of U and dU/dq (exp, log) while
      // the other seven waves add the rows
      double ww = 0.0, ell = 0.0, n = 1.0, n2 = 1.0, lp_wn = 0.0;
      auto head = [&]() {
        ww = __shfl(q, 0);
        ell = __shfl(q, 1);
        n = exp(ell);
        n2 = n * n;
        lp_wn = (-0.5 * ww * ww - AEHMC_LOG_SQRT_2PI) + (log(n) - n + ell);  // lp_w + lp_n of k_linreg_finish
      };
      if (RES && K == 1) {
        if (wave == 0) head();
        else lr_rows_lds<1, LR_BLOCK - 64>(dyn_lds, dyn_lds + a.N, a.N, tid - 64, w4, sxr, srr);
      } else if (RES) {
        lr_rows_lds<K, LR_BLOCK>(dyn_lds, dyn_lds + a.N, a.N, tid, w4, sxr, srr);
      } else {
        lr_rows_direct(a.X, a.y, a.N, wave, lane, w4, sxr, srr);
      }
      if (K == 1) {
        const double s0 = wave_sum(sxr[0]), s1 = wave_sum(srr[0]);
        if (lane == 0) {
          lr_part[wave][0] = s0;
          lr_part[wave][1] = s1;
        }
      } else {  // eight sums in one pass; lane l < 8 ends with the total of value l
        const double v[8] = {sxr[0], srr[0], sxr[1], srr[1], sxr[2], srr[2], sxr[3], srr[3]};
        const double tot = wave_sum8(v, lane);
        if (lane < 8) lr_part[wave][lane] = tot;
      }
      __syncthreads();
      if (!ghost) {
        double s_xr = lr_part[0][2 * wave], s_rr = lr_part[0][2 * wave + 1];
#pragma unroll
        for (int w = 1; w < LR_WAVES; w++) {
          s_xr += lr_part[w][2 * wave];
          s_rr += lr_part[w][2 * wave + 1];
        }
        // U and dU/dq as k_linreg_finish
        if (!(RES && K == 1)) head();
        const double lp_y = -0.5 * (s_rr / n2) - N * AEHMC_LOG_SQRT_2PI - N * ell;
        Unew = -(lp_wn + lp_y);
        g = lane == 0 ? -(-ww + s_xr / n2) : -(2.0 - n - N + s_rr / n2);
        if (ok) p = p - b * g;
      }
    }
    if (!ghost) {
      const double pf = -1.0 * p;  // hmc.py:185
      const double v_fin = vel(pf);
      kd = wave_sum(ok ? v_fin * pf : 0.0);
      if (a.L == 0) Unew = U;
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
        q = qs;
        g = gs;
      }
      if (a.samples && ok) a.samples[((size_t)t * a.C + c) * 2 + lane] = q;
      if (lane == 0) {
        if (a.acc_hist) a.acc_hist[(size_t)t * a.C + c] = pa;
        if (a.div_hist) a.div_hist[(size_t)t * a.C + c] = is_div;
      }
    }
  }
  if (ghost) return;
  if (ok) {
    a.q[c * 2 + lane] = q;
    a.g[c * 2 + lane] = g;
    if (a.out.momentum) a.out.momentum[c * 2 + lane] = acc ? -1.0 * p : p0;
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

inline bool hmc_linreg_supported(int tkind, int met_ndim, long long D) {
  return tkind == AEHMC_T_LINREG && met_ndim <= 2 && D == 2;
}
template <bool RES, int K>
inline hipError_t launch_hmc_linreg_k(const HmcFusedArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)((a.C + 3) / 4)), block(LR_BLOCK);
  const size_t dyn = RES ? (size_t)2 * a.N * sizeof(double) : 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hmc_linreg<RES, K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_hmc_linreg<RES, K>), grid, block, dyn, st, a);
  return hipGetLastError();
}
inline hipError_t launch_hmc_linreg(const HmcFusedArgs &a, hipStream_t st) {
  if (a.N > HMC_LINREG_LDS_ROWS) return launch_hmc_linreg_k<false, 4>(a, st);
  if (a.C == 1) return launch_hmc_linreg_k<true, 1>(a, st);  // the notebook's single chain
  return launch_hmc_linreg_k<true, 4>(a, st);
}

}  // namespace aehmc
