// Fused, register-resident HMC transition (gfx950): one chain per wavefront, the whole
// hmc.new_kernel(...)(state, eps, imm, L) call in ONE launch (k_hmc_fused, D <= 1024); for
// 1024 < D <= 10240 one workgroup per chain with the state in VGPRs, fed by a momentum
// pre-pass (k_hmc_wide, further down).
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
  const double *X, *y;  // regression target: data rows [N]
  long long N;
  uint64_t *rng;  // [C,2,4]
  double *q, *U, *g;
  aehmc_diagnostics out;
  // multi-transition driver (the user-level scan of tests/test_hmc.py:138-148): T transitions
  // per launch, optional per-transition outputs
  long long T;
  double *samples;   // [T,C,D] or null
  double *acc_hist;  // [T,C] or null
  int32_t *div_hist; // [T,C] or null
  int fc;            // "fp_contract" option: fast arithmetic in the leapfrog bodies (1e-6 instead of bit parity)
  const double *const *cparams;  // user-defined target (AEHMC_T_CUSTOM, run-time compiled instantiations only)
};

inline bool target_is_elem_host(int k) {
  return k == AEHMC_T_STD_NORMAL || k == AEHMC_T_ISO_GAUSSIAN || k == AEHMC_T_DIAG_GAUSSIAN;
}
inline bool hmc_fused_supported(int tkind, int met_ndim, long long D) {
  return target_is_elem_host(tkind) && met_ndim < 2 && D <= 1024;
}

// FC ("fp_contract" option, off by default): the leapfrog bodies in fast arithmetic -- every a*b+c one fused
// multiply-add, the loop-invariant products eps*imm and 1/sigma^2 formed once, and the two half kicks that meet
// between consecutive leapfrogs of the static trajectory merged into one full kick (p - eps g instead of
// (p - b g) - b g): 2 fp64 operations per element and leapfrog instead of 6 (diagonal-Gaussian target: 4 instead
// of 8 + two divisions).  Mathematically the integrator of integrators.py:54-73; results within the north star's
// 1e-6 (relative) of the default mode, which stays bit-identical to the oracle.  Everything outside the
// trajectory loop (momentum draw, energies, accept) is the default mode's code.
// TK == AEHMC_T_CUSTOM exists only in the run-time compiled copy (aehmc_set_custom_target): potential and gradient
// of a coordinate come from the user's aehmc_custom_elem
#ifdef AEHMC_CUSTOM_TARGET
#define AEHMC_CUSTOM_ELEM(q, i, u, g) aehmc_custom_elem((q), (i), a.cparams, (u), (g))
#else
#define AEHMC_CUSTOM_ELEM(q, i, u, g) do { (u) = 0.0; (g) = 0.0; } while (0)
#endif
// Wavefronts per SIMD the register allocator leaves room for: 4096 chains are 4 wavefronts per SIMD.  Left alone, the
// contracted-arithmetic variants at R = 4 take 130 / 156 VGPRs (3 wavefronts) and R = 8 174 (2); held to 4 / 3 they fit
// 111 / 127 / 168 with nothing spilled inside the leapfrog loop (the diagonal-Gaussian variant at R = 8 would spill there
// and is left alone).
// (a traced joint density brings registers of its own: no floor)
constexpr int hmc_fused_min_waves(int R, int TK) {
  return TK == AEHMC_T_JOINT ? 1 : R <= 4 ? 4 : (R == 8 && TK != AEHMC_T_DIAG_GAUSSIAN ? 3 : 1);
}

template <int R, int TK, bool FC = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(hmc_fused_min_waves(R, TK)))) void k_hmc_fused(HmcFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double zlds[];  // [4 waves][R*64]
  __shared__ double ztab[ZIG_LDS_DOUBLES];
  const ZigTabLds tab = zig_tab_to_lds(ztab);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c = (long long)blockIdx.x * 4 + w;
  if (c >= a.C) return;
  const size_t row = (size_t)c * a.D;
  // TK == AEHMC_T_JOINT (round 6, run-time compiled copy only): a traced joint density with its reverse-mode gradient
  // (aehmc_logp_grad, tracing.py).  The chain stays in registers as for the coordinate-wise targets; the generated
  // program reads the position from, and adds the gradient into, two rows of the wavefront's LDS (the first doubles as
  // the row of normals): k_hmc_joint_rows without its trips through the L2 between the stages.
  double *zrow = zlds + (size_t)w * ((TK == AEHMC_T_JOINT ? 2 : 1) * R * 64);

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
  double aim[FC ? R : 1], iv[(FC && TK == AEHMC_T_DIAG_GAUSSIAN) ? R : 1];
  (void)iv;
  if (FC) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      aim[FC ? r : 0] = aa * im[r];
      if (TK == AEHMC_T_DIAG_GAUSSIAN) iv[FC ? r : 0] = 1.0 / (sg[r] * sg[r]);
    }
  }

  const PcgLaneJump jump1 = pcg_lane_jump(g1);
  for (long long t = 0; t < a.T; t++) {
    wave_normals(g1, a.D, [=](long long i, double z) { zrow[i] = z; }, tab, jump1);  // metrics.py:65-68
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

    double U_joint = U;
    (void)U_joint;
    if (FC && TK != AEHMC_T_JOINT) {
      constexpr bool DGT = TK == AEHMC_T_DIAG_GAUSSIAN || TK == AEHMC_T_CUSTOM;  // otherwise dU/dq == q
      constexpr bool CUS = TK == AEHMC_T_CUSTOM;
      if (a.L > 0) {
        // half kick | (drift, full kick) x (L - 1) | drift, half kick: no per-iteration select, four leapfrogs per
        // trip of the loop (the body is four instructions per element: loop overhead would otherwise match it)
        const double neg_eps = -eps;
#pragma unroll
        for (int r = 0; r < R; r++) p[r] = __builtin_fma(-b, DGT ? g[r] : q[r], p[r]);
#pragma unroll 4
        for (long long l = 1; l < a.L; l++) {
#pragma unroll
          for (int r = 0; r < R; r++) {
            q[r] = __builtin_fma(aim[FC ? r : 0], p[r], q[r]);
            if (CUS) {
              double u_;
              AEHMC_CUSTOM_ELEM(q[r], (long long)(lane + 64 * r < a.D ? lane + 64 * r : 0), u_, g[r]);
            } else if (DGT) g[r] = (q[r] - mu[r]) * iv[(FC && TK == AEHMC_T_DIAG_GAUSSIAN) ? r : 0];
            p[r] = __builtin_fma(neg_eps, DGT ? g[r] : q[r], p[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
          q[r] = __builtin_fma(aim[FC ? r : 0], p[r], q[r]);
          if (CUS) {
            double u_;
            AEHMC_CUSTOM_ELEM(q[r], (long long)(lane + 64 * r < a.D ? lane + 64 * r : 0), u_, g[r]);
          } else if (DGT) g[r] = (q[r] - mu[r]) * iv[(FC && TK == AEHMC_T_DIAG_GAUSSIAN) ? r : 0];
          p[r] = __builtin_fma(-b, DGT ? g[r] : q[r], p[r]);
        }
      }
      if (!DGT) {
#pragma unroll
        for (int r = 0; r < R; r++) g[r] = q[r];
      }
#ifdef AEHMC_JOINT_GRAD
    } else if (TK == AEHMC_T_JOINT) {
      double *const grow = zrow + R * 64;
      for (long long l = 0; l < a.L; l++) {  // leap_stages<1,1,0>, joint_rows_eval, leap_stages<0,0,1>
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = p[r] - b * g[r];
          q[r] = q[r] + aa * (im[r] * p[r]);
          if (ok[r]) {
            zrow[lane + 64 * r] = q[r];
            grow[lane + 64 * r] = 0.0;
          }
        }
        __threadfence_block();
        const double lp = aehmc_logp_grad(zrow, grow, lane, a.cparams);
        __threadfence_block();
#pragma unroll
        for (int r = 0; r < R; r++) {
          g[r] = ok[r] ? -grow[lane + 64 * r] : 0.0;
          p[r] = p[r] - b * g[r];
        }
        U_joint = -lp;
      }
      __threadfence_block();
#endif
    } else if (TK == AEHMC_T_CUSTOM) {
      for (long long l = 0; l < a.L; l++) {  // leap_stages<1,1,1> with the user's gradient
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = p[r] - b * g[r];
          q[r] = q[r] + aa * (im[r] * p[r]);
          double u_;
          AEHMC_CUSTOM_ELEM(q[r], (long long)(lane + 64 * r < a.D ? lane + 64 * r : 0), u_, g[r]);
          p[r] = p[r] - b * g[r];
        }
      }
    } else if (TK == AEHMC_T_DIAG_GAUSSIAN) {
      for (long long l = 0; l < a.L; l++) {  // trajectory.py:86-95, integrators.py:54-73
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = p[r] - b * g[r];
          q[r] = q[r] + aa * (im[r] * p[r]);
          g[r] = ((q[r] - mu[r]) / sg[r]) / sg[r];
          p[r] = p[r] - b * g[r];
        }
      }
    } else {
      // dU/dq == q: no separate gradient registers in the loop, and the product b * q' that ends one
      // leapfrog is the same number that starts the next: 6 fp64 operations per element and step
      double bq[R];
#pragma unroll
      for (int r = 0; r < R; r++) bq[r] = b * q[r];
      for (long long l = 0; l < a.L; l++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = p[r] - bq[r];
          q[r] = q[r] + aa * (im[r] * p[r]);
          bq[r] = b * q[r];
          p[r] = p[r] - bq[r];
        }
      }
#pragma unroll
      for (int r = 0; r < R; r++) g[r] = q[r];
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
        else if (TK == AEHMC_T_JOINT) {
        } else if (TK == AEHMC_T_CUSTOM) {
          double u_, g_;
          AEHMC_CUSTOM_ELEM(q[r], i, u_, g_);
          usum += u_;
        } else {
          double z = (q[r] - mu[r]) / sg[r];
          usum += 0.5 * (z * z) + a.log_sigma[i] + AEHMC_LOG_SQRT_2PI;
        }
        double pf = -1.0 * p[r];  // hmc.py:185
        kd += (im[r] * pf) * pf;
      }
    }
    usum = wave_sum(usum);
    kd = wave_sum(kd);
    const double Unew = TK == AEHMC_T_JOINT ? U_joint : a.L > 0 ? (TK == AEHMC_T_ISO_GAUSSIAN ? 0.5 * usum : usum) : U;
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

// ---- large D: one workgroup of T threads per chain, state in VGPRs ----------------------
// Thread t keeps elements t, t+T, ... of q, p (and dU/dq where it is not q itself) in registers for
// all L leapfrogs of `nt` CONSECUTIVE transitions (round 3; one per launch before): the chain's position
// never leaves the chip between the transitions of a sample() call.  The momenta of the nt transitions
// were drawn by k_draw_momentum (one wavefront per chain: the PCG64 stream of a chain is sequential) into
// zbuf[nt][C][D].  Per transition HBM sees the D normals on the way in (loaded into the registers of the
// dead momentum behind the last energy sum, i.e. under the accept arithmetic) and nothing on the way out
// but the optional sample row; q, dU/dq, U go out once, at the end of the launch, if any transition was
// accepted.  The position a rejection falls back to waits in LDS (each thread re-reads only what it wrote
// itself: no barrier).  Diagonal-Gaussian target: the LDS holds sigma and mu instead (two divisions by sigma per
// element and leapfrog: from L2 they cost the loop its registers -- 264 B/lane of scratch, 5.6 ms per transition
// at D = 1e4), and the fall-back state is the caller's q / dU/dq, rewritten at every accepted transition.  Cross-wave sums take one LDS hop, so the summation order differs from the lock-step
// path (1e-13).  Slots past D replicate element D-1 (in bounds), are masked out of the sums, never stored.
template <int T, int R, int TK, bool FC = false>
__global__ __launch_bounds__(T) void k_hmc_wide(HmcFusedArgs a, const double *zbuf, int nt) {
  constexpr int NW = T / 64;
  // TK == AEHMC_T_CUSTOM exists only in the run-time compiled copy (aehmc_set_custom_target, round 5): the user's
  // aehmc_custom_elem; dU/dq kept beside q and the caller's arrays as the fall-back state, as for the diagonal
  // Gaussian, but no parameters of the engine's own in LDS
  constexpr bool CU = TK == AEHMC_T_CUSTOM;
  constexpr bool DG = TK == AEHMC_T_DIAG_GAUSSIAN || CU;  // otherwise dU/dq == q, no separate copy
  constexpr bool DGP = DG && !CU;                          // sigma / mu in LDS
  __shared__ double red[2][2 * NW];
  extern __shared__ __attribute__((aligned(16))) double wide_save[];  // q [D] at the transition's start; DG: sigma [D], mu [D]
  double *const psig = wide_save, *const pmu = wide_save + a.D;       // (diagonal-Gaussian target only)
  int flip = 0;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long long c = blockIdx.x;
  const size_t row = (size_t)c * a.D;
  const size_t imo = (size_t)c * a.imm_cs;
  const size_t tstride = (size_t)a.C * a.D;  // between the rows of one chain in [nt][C][D] arrays
  const unsigned last = (unsigned)a.D - 1;
  // Element indices are derived from `tb`, a copy of the thread index behind an opaque barrier that is renewed at
  // every phase of every transition: otherwise the ~60 addresses of the phases outside the leapfrog loop are
  // hoisted out of the transition loop, stay live across the leapfrog loop next to the 80 registers of q, p, b q
  // and imm, and spill (228 B/lane at <1024, 10>, the spill code inside the leapfrog loop).
  unsigned tb = (unsigned)t;
#define RENEW_TB() asm volatile("" : "+v"(tb))
#define EI(r) (((tb + T * (r)) < last) ? (tb + T * (r)) : last)
#define VALID(r) ((tb + T * (r)) <= last)
#define MASK(r) (VALID(r) ? 1.0 : 0.0)
  auto sum2 = [&](double &x, double &y) {
    x = wave_sum(x);
    y = wave_sum(y);
    double *buf = red[flip];
    flip ^= 1;
    if (lane == 0) {
      buf[2 * wave] = x;
      buf[2 * wave + 1] = y;
    }
    __syncthreads();
    double sx = buf[0], sy = buf[1];
#pragma unroll
    for (int w = 1; w < NW; w++) {
      sx += buf[2 * w];
      sy += buf[2 * w + 1];
    }
    x = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sx)),
                         __builtin_amdgcn_readfirstlane(__double2loint(sx)));
    y = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sy)),
                         __builtin_amdgcn_readfirstlane(__double2loint(sy)));
  };
  double q[R], p[R], g[DG ? R : 1], im[R];
  const double *qrow = a.q + row, *grow = a.g + row, *zrow = zbuf + row;
#pragma unroll
  for (int r = 0; r < R; r++) {  // every load is issued before the first use
    q[r] = qrow[EI(r)];
    p[r] = zrow[EI(r)];
    if (DG) g[r] = grow[EI(r)];
    im[r] = a.imm[imo + (a.met_ndim == 0 ? 0 : EI(r))];
    if (DGP) {  // (a thread reads back only the entries it wrote: no barrier)
      const double sd0 = a.sigma[EI(r)];
      psig[EI(r)] = FC ? 1.0 / (sd0 * sd0) : sd0;  // FC: the reciprocal variance, formed once
      pmu[EI(r)] = a.mu[EI(r)];
    }
  }
#define GR(r) (DG ? g[DG ? (r) : 0] : q[r])
  double U = a.U[c];
  Pcg64 g2 = pcg_load(a.rng + (size_t)c * 8 + 4);  // site #2: accept (hmc.py:194)
  const double eps = a.eps_c ? a.eps_c[c] : a.eps;
  const double b = 0.5 * eps, aa = 1 * eps;
  double pa = 0.0;
  int is_div = 0, acc = 0, any_acc = 0;
  for (int tt = 0; tt < nt; tt++) {
    const bool last_t = tt == nt - 1;
    double kd = 0.0, zero = 0.0;
    RENEW_TB();
#pragma unroll
    for (int r = 0; r < R; r++) {
      kd += MASK(r) * ((im[r] * p[r]) * p[r]);
      if (!DG) wide_save[EI(r)] = q[r];
      // only the last transition's momentum is observable: the initial one is kept on rejection
      if (last_t && a.out.momentum && VALID(r)) (a.out.momentum + row)[EI(r)] = p[r];
    }
    sum2(kd, zero);
    const double H0 = U + 0.5 * kd;  // hmc.py:187
    if (FC) {  // fast arithmetic (see k_hmc_fused): fused multiply-adds, eps * imm formed once, inner half kicks merged
      if (a.L > 0) {
        double aim[FC ? R : 1];
#pragma unroll
        for (int r = 0; r < R; r++) {
          aim[FC ? r : 0] = aa * im[r];
          p[r] = __builtin_fma(-b, GR(r), p[r]);
        }
        const double neg_eps = -eps;
#pragma unroll 4
        for (long long l = 1; l < a.L; l++) {  // (drift, full kick) x (L - 1)
          if (DG) RENEW_TB();
#pragma unroll
          for (int r = 0; r < R; r++) {
            q[r] = __builtin_fma(aim[FC ? r : 0], p[r], q[r]);
            if (CU) {
              double u_;
              AEHMC_CUSTOM_ELEM(q[r], (long long)EI(r), u_, g[DG ? r : 0]);
            } else if (DG) {
              g[DG ? r : 0] = (q[r] - pmu[EI(r)]) * psig[EI(r)];  // psig holds 1 / sigma^2 in this mode
            }
            p[r] = __builtin_fma(neg_eps, GR(r), p[r]);
          }
        }
        if (DG) RENEW_TB();
#pragma unroll
        for (int r = 0; r < R; r++) {  // last drift, half kick
          q[r] = __builtin_fma(aim[FC ? r : 0], p[r], q[r]);
          if (CU) {
            double u_;
            AEHMC_CUSTOM_ELEM(q[r], (long long)EI(r), u_, g[DG ? r : 0]);
          } else if (DG) {
            g[DG ? r : 0] = (q[r] - pmu[EI(r)]) * psig[EI(r)];
          }
          p[r] = __builtin_fma(-b, GR(r), p[r]);
        }
      }
    } else if (DG) {
      for (long long l = 0; l < a.L; l++) {  // trajectory.py:86-95, integrators.py:54-73
        RENEW_TB();
#pragma unroll
        for (int r = 0; r < R; r++) {
          double pp = p[r] - b * GR(r);
          const double qq = q[r] + aa * (im[r] * pp);
          double gg;
          if (CU) {
            double u_;
            AEHMC_CUSTOM_ELEM(qq, (long long)EI(r), u_, gg);
          } else {
            const double sd = psig[EI(r)];
            gg = ((qq - pmu[EI(r)]) / sd) / sd;
          }
          pp = pp - b * gg;
          q[r] = qq;
          g[DG ? r : 0] = gg;
          p[r] = pp;
        }
      }
    } else {  // dU/dq == q; b * q' ends one leapfrog and starts the next (same product, computed once)
      double bq[R];
#pragma unroll
      for (int r = 0; r < R; r++) bq[r] = b * q[r];
      for (long long l = 0; l < a.L; l++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = p[r] - bq[r];
          q[r] = q[r] + aa * (im[r] * p[r]);
          bq[r] = b * q[r];
          p[r] = p[r] - bq[r];
        }
      }
    }
    double usum = 0.0;
    kd = 0.0;
    RENEW_TB();
#pragma unroll
    for (int r = 0; r < R; r++) {
      const double qq = q[r];
      double u;
      if (TK == AEHMC_T_STD_NORMAL) u = 0.5 * (qq * qq) + AEHMC_LOG_SQRT_2PI;
      else if (TK == AEHMC_T_ISO_GAUSSIAN) u = qq * qq;
      else if (CU) {  // (the potential is needed at the trajectory's end only: one more evaluation per transition)
        double g_;
        AEHMC_CUSTOM_ELEM(qq, (long long)EI(r), u, g_);
      } else if (FC) {  // z^2 = d^2 / sigma^2 with the reciprocal variance in LDS
        const double d = qq - pmu[EI(r)];
        u = 0.5 * ((d * d) * psig[EI(r)]) + a.log_sigma[EI(r)] + AEHMC_LOG_SQRT_2PI;
      } else {
        const double z = (qq - pmu[EI(r)]) / psig[EI(r)];
        u = 0.5 * (z * z) + a.log_sigma[EI(r)] + AEHMC_LOG_SQRT_2PI;
      }
      usum += MASK(r) * u;
      const double pf = -1.0 * p[r];  // hmc.py:185
      kd += MASK(r) * ((im[r] * pf) * pf);
    }
    RENEW_TB();
    if (!last_t) {  // the momentum is dead: its registers take the next transition's normals, which arrive
      //               while the sums below are reduced and the accept decision is made
      const double *zn = zrow + (size_t)(tt + 1) * tstride;
#pragma unroll
      for (int r = 0; r < R; r++) p[r] = zn[EI(r)];
    }
    sum2(usum, kd);
    const double Unew = a.L > 0 ? (TK == AEHMC_T_ISO_GAUSSIAN ? 0.5 * usum : usum) : U;
    double delta = H0 - (Unew + 0.5 * kd);
    if (isnan(delta)) delta = -INFINITY;
    is_div = fabs(delta) > a.thr;
    pa = exp(delta);
    if (pa > 1.0) pa = 1.0;
    if (pa < 0.0) pa = 0.0;
    acc = rng_bernoulli(g2, pa);  // hmc.py:193-195
    RENEW_TB();
    if (acc) {
      U = Unew;
      any_acc = 1;
      if (DG) {  // the caller's arrays are this target's fall-back state: they follow every accepted transition
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (!VALID(r)) continue;
          (a.q + row)[EI(r)] = q[r];
          (a.g + row)[EI(r)] = g[DG ? r : 0];
        }
      }
    } else {  // back to the transition's start
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (DG) {
          q[r] = qrow[EI(r)];
          g[DG ? r : 0] = grow[EI(r)];
        } else {
          q[r] = wide_save[EI(r)];
        }
      }
    }
    if (a.samples) {
      double *dst = a.samples + (size_t)tt * tstride + row;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (VALID(r)) dst[EI(r)] = q[r];
    }
    if (t == 0) {
      if (a.acc_hist) a.acc_hist[(size_t)tt * a.C + c] = pa;
      if (a.div_hist) a.div_hist[(size_t)tt * a.C + c] = is_div;
    }
  }
  RENEW_TB();
  if (any_acc && !DG) {  // commit; a call without an accepted transition leaves the state in HBM untouched
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!VALID(r)) continue;
      (a.q + row)[EI(r)] = q[r];
      (a.g + row)[EI(r)] = GR(r);
    }
  }
  if (acc && a.out.momentum) {
#pragma unroll
    for (int r = 0; r < R; r++)
      if (VALID(r)) (a.out.momentum + row)[EI(r)] = -1.0 * p[r];
  }
  if (t == 0) {
    pcg_store(a.rng + (size_t)c * 8 + 4, g2);
    if (any_acc) a.U[c] = U;
    a.out.acceptance_probability[c] = pa;
    a.out.is_diverging[c] = is_div;
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = a.L * nt;
    if (a.out.is_turning) a.out.is_turning[c] = acc;  // HMC: reused as the accept flag
  }
#undef GR
#undef EI
#undef VALID
#undef MASK
#undef RENEW_TB
}

inline bool hmc_resident_supported(int tkind, int met_ndim, long long D) {
  // (the diagonal-Gaussian target parks q AND dU/dq in LDS between a transition's start and its accept
  //  decision: 16 D bytes next to the reduction scratch in the CU's 160 KB)
  return target_is_elem_host(tkind) && met_ndim < 2 && D > 1024 && D <= (tkind == AEHMC_T_DIAG_GAUSSIAN ? 10176 : 10240);
}
#ifndef __HIPCC_RTC__
template <int T, int R>
inline hipError_t launch_hmc_wide_r(const HmcFusedArgs &a, const double *zbuf, int nt, hipStream_t st) {
  const bool dg = a.tkind == AEHMC_T_DIAG_GAUSSIAN;
  const size_t dyn = (size_t)a.D * sizeof(double) * (dg ? 2 : 1);
#define AEHMC_WIDE_LAUNCH_FC(TK, FCV)                                                                      \
  do {                                                                                                     \
    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hmc_wide<T, R, TK, FCV>),        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);             \
    if (e_ != hipSuccess) return e_;                                                                       \
    hipLaunchKernelGGL((k_hmc_wide<T, R, TK, FCV>), dim3((unsigned)a.C), dim3(T), dyn, st, a, zbuf, nt);   \
  } while (0)
#define AEHMC_WIDE_LAUNCH(TK)                  \
  do {                                         \
    if (a.fc) AEHMC_WIDE_LAUNCH_FC(TK, true);  \
    else AEHMC_WIDE_LAUNCH_FC(TK, false);      \
  } while (0)
  switch (a.tkind) {
    case AEHMC_T_STD_NORMAL:
      AEHMC_WIDE_LAUNCH(AEHMC_T_STD_NORMAL);
      break;
    case AEHMC_T_ISO_GAUSSIAN:
      AEHMC_WIDE_LAUNCH(AEHMC_T_ISO_GAUSSIAN);
      break;
    default:
      AEHMC_WIDE_LAUNCH(AEHMC_T_DIAG_GAUSSIAN);
  }
#undef AEHMC_WIDE_LAUNCH
#undef AEHMC_WIDE_LAUNCH_FC
  return hipGetLastError();
}
// `nt` consecutive transitions; `samples`, `acc_hist`, `div_hist` point at the FIRST of them ([nt][C][..] slices);
// zbuf [nt][C][D] holds their momenta.
inline hipError_t launch_hmc_resident(const HmcFusedArgs &a, const double *zbuf, int nt, hipStream_t st) {
  if (a.D <= 2048) return launch_hmc_wide_r<256, 8>(a, zbuf, nt, st);
  if (a.D <= 4096) return launch_hmc_wide_r<512, 8>(a, zbuf, nt, st);
  if (a.D <= 8192) return launch_hmc_wide_r<1024, 8>(a, zbuf, nt, st);
  return launch_hmc_wide_r<1024, 10>(a, zbuf, nt, st);
}

template <int R, bool FC>
inline hipError_t launch_hmc_fused_r(const HmcFusedArgs &a, hipStream_t st) {
  dim3 grid((unsigned)((a.C + 3) / 4)), block(256);
  size_t lds = (size_t)4 * R * 64 * sizeof(double);
  switch (a.tkind) {
    case AEHMC_T_STD_NORMAL:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_STD_NORMAL, FC>), grid, block, lds, st, a);
      break;
    case AEHMC_T_ISO_GAUSSIAN:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_ISO_GAUSSIAN, FC>), grid, block, lds, st, a);
      break;
    default:
      hipLaunchKernelGGL((k_hmc_fused<R, AEHMC_T_DIAG_GAUSSIAN, FC>), grid, block, lds, st, a);
  }
  return hipGetLastError();
}
template <bool FC>
inline hipError_t launch_hmc_fused_fc(const HmcFusedArgs &a, hipStream_t st) {
  const long long r = (a.D + 63) / 64;
  if (r <= 1) return launch_hmc_fused_r<1, FC>(a, st);
  if (r <= 2) return launch_hmc_fused_r<2, FC>(a, st);
  if (r <= 4) return launch_hmc_fused_r<4, FC>(a, st);
  if (r <= 8) return launch_hmc_fused_r<8, FC>(a, st);
  return launch_hmc_fused_r<16, FC>(a, st);
}
inline hipError_t launch_hmc_fused(const HmcFusedArgs &a, hipStream_t st) {
  return a.fc ? launch_hmc_fused_fc<true>(a, st) : launch_hmc_fused_fc<false>(a, st);
}
#endif  // __HIPCC_RTC__
// elements per lane of the k_hmc_fused instantiation that holds a D-dimensional chain
inline int hmc_fused_r(long long D) {
  const long long r = (D + 63) / 64;
  return r <= 1 ? 1 : r <= 2 ? 2 : r <= 4 ? 4 : r <= 8 ? 8 : 16;
}

}  // namespace aehmc
