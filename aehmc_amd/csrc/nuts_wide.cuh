// Workgroup-per-chain NUTS transition for large D (512 < D <= 10176), gfx950.
//
// The whole nuts.new_kernel(...)(state, eps, imm) call of ONE chain runs in one workgroup of T
// threads (T = 256 / 512); the chain's moving end lives on chip for the entire tree:
// thread t owns elements t, t+T, ... -- p and the sub-trajectory momentum sum in VGPRs, q (and
// dU/dq where it is not q itself) in VGPRs or, for D > 4096, in LDS (160 KB per CU = one
// D = 1e4 chain).  Diagonal / scalar metric (shared or per chain), coordinate-wise targets.
//
// What makes a leapfrog cheap here (round 2; the round-1 version of this kernel ran at 26 us
// per leapfrog and CU, 1 KB/lane of scratch):
//  * ONE pass over the thread's elements per leapfrog does the whole integrator step
//    (integrators.py:54-73), the kinetic energy (metrics.py:70-73), the running momentum sum
//    and checkpoint stores (termination.py:115-124) AND the first level of the iterative
//    U-turn check (termination.py:133-187), followed by ONE 4-value team reduction (one barrier).
//  * The first level of that check needs no memory: at an odd step the checkpoint of level
//    idx_max is the one the previous (even) step stored -- its momentum is this step's p
//    BEFORE the update and its momentum sum is the running sum BEFORE the update, both still in
//    registers.  (Holds whenever idx_max(step) is the index the previous step stored to; the
//    stale step-0 indices of termination.py:109-113 are the one exception and are checked.)
//    Deeper levels (on average 1/2 per leapfrog) are read from the checkpoint arrays.
//  * Per-element parameters (imm; mu, sigma, log sigma of a diagonal target) are L2-resident
//    vectors shared by all chains: kept in VGPRs when they fit (R <= 8), otherwise streamed in
//    batches with the next batch's loads issued before the current batch's arithmetic.
// HBM then sees per leapfrog: the checkpoint pair of every other step (8 D bytes on
// average), ~1/2 checkpoint pair read (8 D), the proposal copy on accept and the
// trajectory ends at expansion boundaries -- not the 88 D bytes of a streaming step.
//
// Arithmetic per element is that of engine.cuh's lock-step path; sums are accumulated per
// thread in ascending element order, then wave (DPP) and cross-wave in a fixed order, so
// results agree with the oracle to rounding (tested at 1e-9), independent of the launch.
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235,
// proposals.py:19-174, integrators.py:54-73, metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"

namespace aehmc {

template <int T, int R, bool QGL, int TK>
__global__ __launch_bounds__(T) void k_nuts_wide(EngineArgs a) {
  constexpr int NW = T / 64;
  constexpr bool DG = TK == AEHMC_T_DIAG_GAUSSIAN;  // otherwise dU/dq == q: no separate copy
  constexpr bool PAR_REG = R <= 8;                  // per-element parameters live in VGPRs
  constexpr int BR = (R % 4 == 0) ? 4 : R;  // elements per streamed batch
  constexpr int NB = R / BR;
  static_assert(R % BR == 0, "R must be a multiple of the batch size");
  __shared__ double red[2][4 * NW];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  double *const sq = dyn_lds, *const sg = dyn_lds + (QGL && DG ? a.D : 0);
  int flip = 0;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long long c = blockIdx.x;
  const size_t row = (size_t)c * a.D;
  const unsigned last = (unsigned)a.D - 1;
  const bool im_scalar = a.met_ndim == 0;
  const double *const imrow = a.imm + (size_t)c * a.imm_cs;
  // Thread t owns slots r = 0 .. R-1 = elements t + T r.  Slots r < nfull are valid in every thread
  // (no predicate in their code), slot nfull is valid for t < D - T nfull, later slots are never
  // touched -- nfull is wave-uniform, so this costs one scalar branch per slot.
  const int nfull = (int)(a.D / T), nslots = (int)((a.D + T - 1) / T);
  // `tt` is t behind an opaque barrier that is renewed in every loop iteration: element
  // addresses derived from it cannot be hoisted out of the loops (the compiler would otherwise
  // precompute a 64-bit address per element and array -- hundreds of VGPRs -- and spill them)
  int tt = t;
#define AEHMC_FRESH_TT() asm volatile("" : "+v"(tt))
// loads use the clamped index (slots past D re-read element D-1: in bounds, masked out of every
// sum, never stored)
#define EI(r) (((unsigned)(tt + T * (r)) < last) ? (unsigned)(tt + T * (r)) : last)
#define VALID(r) ((unsigned)(tt + T * (r)) <= last)
#define AT(ptr, r) ((ptr) + row)[EI(r)]
// body for slot r: unpredicated when the slot is full, predicated by `on` for the ragged slot
#define AEHMC_SLOT(r, ...)                   \
  do {                                       \
    if ((r) < nfull) {                       \
      constexpr bool on = true;              \
      (void)on;                              \
      __VA_ARGS__                            \
    } else if ((r) < nslots) {               \
      const bool on = VALID(r);              \
      __VA_ARGS__                            \
    }                                        \
  } while (0)

  // team sum of four values; every thread returns the same bits (SGPRs)
  auto sum4 = [&](double &x0, double &x1, double &x2, double &x3) {
    x0 = wave_sum(x0);
    x1 = wave_sum(x1);
    x2 = wave_sum(x2);
    x3 = wave_sum(x3);
    double *buf = red[flip];
    flip ^= 1;  // double-buffered: the next reduction writes the other buffer
    if (lane == 0) {
      buf[4 * wave] = x0;
      buf[4 * wave + 1] = x1;
      buf[4 * wave + 2] = x2;
      buf[4 * wave + 3] = x3;
    }
    __syncthreads();
    double s0 = buf[0], s1 = buf[1], s2 = buf[2], s3 = buf[3];
#pragma unroll
    for (int w = 1; w < NW; w++) {
      s0 += buf[4 * w];
      s1 += buf[4 * w + 1];
      s2 += buf[4 * w + 2];
      s3 += buf[4 * w + 3];
    }
    auto uni = [](double v) {
      return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                              __builtin_amdgcn_readfirstlane(__double2loint(v)));
    };
    x0 = uni(s0);
    x1 = uni(s1);
    x2 = uni(s2);
    x3 = uni(s3);
  };

  // per-element parameters of one element
  struct Par {
    double im, mu, sd, ls;
  };
  auto par_load = [&](int r) {
    Par x;
    const unsigned i = EI(r);
    x.im = imrow[im_scalar ? 0u : i];
    x.mu = DG ? a.mu[i] : 0.0;
    x.sd = DG ? a.sigma[i] : 1.0;
    x.ls = DG ? a.log_sigma[i] : 0.0;
    return x;
  };

  double q[QGL ? 1 : R], g[(QGL || !DG) ? 1 : R], p[R], pb[R];
  Par preg[PAR_REG ? R : 1];
// (QGL: a thread only ever touches the LDS slots of its own valid elements -- no barriers)
#define QGET(r) (QGL ? sq[EI(r)] : q[QGL ? 0 : (r)])
#define QSET(r, v) do { if (QGL) { if (on) sq[EI(r)] = (v); } else q[QGL ? 0 : (r)] = (v); } while (0)
#define GGET(r) (!DG ? QGET(r) : (QGL ? sg[EI(r)] : g[(QGL || !DG) ? 0 : (r)]))
#define GSET(r, v) do { if (DG) { if (QGL) { if (on) sg[EI(r)] = (v); } else g[(QGL || !DG) ? 0 : (r)] = (v); } } while (0)
#define IMOF(r) (PAR_REG ? preg[PAR_REG ? (r) : 0].im : imrow[im_scalar ? 0u : EI(r)])

  // ---- load the chain: q, dU/dq, momentum (drawn by k_draw_momentum into zbuf), parameters;
  //      nuts.py:113-125 --------------------------------------------------------------------
  double kd = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    p[r] = pb[r] = 0.0;
    if (!QGL) q[QGL ? 0 : r] = 0.0;
    if (!QGL && DG) g[(QGL || !DG) ? 0 : r] = 0.0;
    if (PAR_REG) preg[PAR_REG ? r : 0] = Par{1.0, 0.0, 1.0, 0.0};
    AEHMC_SLOT(r, {
      const double qv = AT(a.q, r), gv = DG ? AT(a.g, r) : qv, pv = AT(a.zbuf, r);
      if (PAR_REG) preg[PAR_REG ? r : 0] = par_load(r);
      const double im = IMOF(r);
      p[r] = pv;
      QSET(r, qv);
      GSET(r, gv);
      if (on) {
        kd += (im * pv) * pv;
        AT(a.end_q[0], r) = qv;
        AT(a.end_p[0], r) = pv;
        AT(a.end_g[0], r) = gv;
        AT(a.end_q[1], r) = qv;
        AT(a.end_p[1], r) = pv;
        AT(a.end_g[1], r) = gv;
        AT(a.slot_q[0], r) = qv;
        AT(a.slot_p[0], r) = pv;
        AT(a.slot_g[0], r) = gv;
        AT(a.psum, r) = pv;
      }
    });
  }
  ChainRng rng = rng_load(a, c);
  ChainCtl ct = {};
  sum4(kd, z1, z2, z3);
  {
    const double U = a.U[c];
    ct.H0 = U + 0.5 * kd;
    ct.prop_E = ct.H0;
    ct.prop_w = 0.0;
    ct.prop_slpa = -INFINITY;
    ct.U_cur = ct.U_end[0] = ct.U_end[1] = ct.U_slot[0] = ct.U_slot[1] = U;
    ct.dir = rng_bernoulli(rng.g[1], 0.5);  // trajectory.py:516
  }
  const double eps = a.eps_c ? a.eps_c[c] : a.eps;
  int ck_last = -1;  // checkpoint index the previous step stored to (-1: none)

  while (!ct.done) {
    AEHMC_FRESH_TT();
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size, aa = 1 * step_size;
    const int step = ct.step;
    if (!ct.phantom) ct.nleap += 1;
    int tmin, tmax;
    if (step == 0) {
      tmin = ct.tmin;  // termination.py:109-113: stale indices of the previous sub-trajectory
      tmax = ct.tmax;
    } else {
      const int n1 = __ffs(~step) - 1;
      tmax = __popc(step >> 1);
      tmin = tmax - n1 + 1;
    }
    const bool even = (step & 1) == 0;
    const bool check = step >= 1 && tmax >= tmin;
    // level tmax of the check is the pair the previous step stored: p and the momentum sum as
    // they are in registers before this step's update
    const bool fwd = check && ck_last == tmax;
    const double fwd_m = fwd ? 1.0 : 0.0;
    double *const ckp = a.ckp + ((size_t)tmax * a.C + c) * a.D;
    double *const cks = a.cks + ((size_t)tmax * a.C + c) * a.D;

    // ---- one pass: leapfrog + kinetic energy + momentum sum + checkpoint + first U-turn level ----
    double usum = 0.0, d_l = 0.0, d_r = 0.0;
    kd = 0.0;
    Par cur[BR], nxt[BR];
    if (!PAR_REG) {
#pragma unroll
      for (int u = 0; u < BR; u++) cur[u] = par_load(u);
    }
#pragma unroll
    for (int b0 = 0; b0 < NB; b0++) {
      if (!PAR_REG && b0 + 1 < NB) {
#pragma unroll
        for (int u = 0; u < BR; u++) nxt[u] = par_load((b0 + 1) * BR + u);
      }
#pragma unroll
      for (int u = 0; u < BR; u++) {
        const int r = b0 * BR + u;
        const Par x = PAR_REG ? preg[PAR_REG ? r : 0] : cur[u];
        AEHMC_SLOT(r, {
          const double p_old = p[r], pb_old = pb[r];
          double pp = p_old - b * GGET(r);               // integrators.py:59-60
          const double qq = QGET(r) + aa * (x.im * pp);  // integrators.py:62-64
          double uu, gg;
          if (TK == AEHMC_T_STD_NORMAL) {
            uu = 0.5 * (qq * qq) + AEHMC_LOG_SQRT_2PI;
            gg = qq;
          } else if (TK == AEHMC_T_ISO_GAUSSIAN) {
            uu = qq * qq;
            gg = qq;
          } else {
            const double z = (qq - x.mu) / x.sd;
            uu = 0.5 * (z * z) + x.ls + AEHMC_LOG_SQRT_2PI;
            gg = z / x.sd;
          }
          pp = pp - b * gg;                               // integrators.py:67-69
          QSET(r, qq);
          GSET(r, gg);
          p[r] = pp;
          const double v = x.im * pp;
          const double s = (step == 0) ? pp : pb_old + pp;  // trajectory.py:278,243
          pb[r] = s;
          // termination.py:160-173 for level idx_max, from registers (weight 0 when not due)
          const double pl = p_old, vl = x.im * pl;
          const double sub = s - pb_old + pl;
          const double rho = sub - (pp + pl) / 2;
          usum += on ? uu : 0.0;
          kd += on ? v * pp : 0.0;
          d_l += on ? fwd_m * (vl * rho) : 0.0;
          d_r += on ? fwd_m * (v * rho) : 0.0;
          if (even && on) {  // termination.py:115-124
            ckp[EI(r)] = pp;
            cks[EI(r)] = s;
          }
        });
      }
      if (!PAR_REG && b0 + 1 < NB) {
#pragma unroll
        for (int u = 0; u < BR; u++) cur[u] = nxt[u];
      }
      if (R > 8) __builtin_amdgcn_sched_barrier(0);
    }
    if (even) ck_last = tmax;
    sum4(usum, kd, d_l, d_r);
    ct.U_cur = (TK == AEHMC_T_ISO_GAUSSIAN) ? 0.5 * usum : usum;
    ct.tmin = tmin;
    ct.tmax = tmax;

    // ---- dynamic_integration body (trajectory.py:195-305), per-chain scalars ------------------
    const double E = ct.U_cur + 0.5 * kd;  // proposals.py:19-62
    double delta = ct.H0 - E;
    if (isnan(delta)) delta = -INFINITY;
    const bool div = fabs(delta) > a.thr;
    const double np_w = delta, np_slpa = delta > 0 ? 0.0 : delta;
    bool term = false, take = false;
    if (step == 0) {
      ct.sub_E = E;
      ct.sub_w = np_w;
      ct.sub_slpa = np_slpa;
      ct.length = 1;
      take = true;
    } else {
      double pa = 1.0 / (1.0 + exp(-(np_w - ct.sub_w)));  // proposals.py:96-99
      if (isnan(pa)) pa = 0.0;
      const int acc = rng_bernoulli(rng.g[2], pa);
      ct.sub_w = np_logaddexp(ct.sub_w, np_w);
      ct.sub_slpa = np_logaddexp(ct.sub_slpa, np_slpa);
      if (acc) {
        ct.sub_E = E;
        take = !ct.phantom;
      }
      ct.length += 1;
      if (check) {  // termination.py:133-187
        int idx = tmax;
        bool crit = false;
        for (;;) {
          if (!(fwd && idx == tmax)) {  // a level that is not in registers: read its checkpoint pair
            AEHMC_FRESH_TT();
            const double *kp = a.ckp + ((size_t)idx * a.C + c) * a.D;
            const double *ks = a.cks + ((size_t)idx * a.C + c) * a.D;
            d_l = 0.0;
            d_r = 0.0;
            double kpc[BR], ksc[BR], imc[BR], kpn[BR], ksn[BR], imn[BR];
#pragma unroll
            for (int u = 0; u < BR; u++) {
              kpc[u] = kp[EI(u)];
              ksc[u] = ks[EI(u)];
              imc[u] = IMOF(u);
            }
#pragma unroll
            for (int b0 = 0; b0 < NB; b0++) {
              if (b0 + 1 < NB) {
#pragma unroll
                for (int u = 0; u < BR; u++) {
                  const int r = (b0 + 1) * BR + u;
                  kpn[u] = kp[EI(r)];
                  ksn[u] = ks[EI(r)];
                  imn[u] = IMOF(r);
                }
              }
#pragma unroll
              for (int u = 0; u < BR; u++) {
                const int r = b0 * BR + u;
                AEHMC_SLOT(r, {
                  const double pl = kpc[u], pr = p[r];
                  const double vl = imc[u] * pl, vr = imc[u] * pr;
                  const double sub = pb[r] - ksc[u] + pl;
                  const double rho = sub - (pr + pl) / 2;
                  d_l += on ? vl * rho : 0.0;
                  d_r += on ? vr * rho : 0.0;
                });
              }
              if (b0 + 1 < NB) {
#pragma unroll
                for (int u = 0; u < BR; u++) {
                  kpc[u] = kpn[u];
                  ksc[u] = ksn[u];
                  imc[u] = imn[u];
                }
              }
              if (R > 8) __builtin_amdgcn_sched_barrier(0);
            }
            double e0 = 0.0, e1 = 0.0;
            sum4(d_l, d_r, e0, e1);
          }
          crit = (d_l <= 0) | (d_r <= 0);
          const bool reached = (idx - 1) < tmin;
          idx -= 1;
          if (crit || reached) break;
        }
        term = crit;
      }
    }
    if (take) {  // sub-trajectory proposal <- moving end (copy on accept)
      AEHMC_FRESH_TT();
      const int s = ct.prop_slot ^ 1;
#pragma unroll
      for (int r = 0; r < R; r++) {
        AEHMC_SLOT(r, {
          if (on) {
            AT(pick2(a.slot_q, s), r) = QGET(r);
            AT(pick2(a.slot_p, s), r) = p[r];
            AT(pick2(a.slot_g, s), r) = GGET(r);
          }
        });
      }
      put2(ct.U_slot, s, ct.U_cur);
    }

    // ---- sub-trajectory / expansion control (trajectory.py:336, 537-608) ----------------------
    bool finalize = false, fin_div = false, fin_term = false;
    if (step == 0 && div && !ct.phantom) {
      finalize = true;
      fin_div = true;
    } else if (step >= 1 && (div || term || step == (1 << ct.j))) {
      if (ct.phantom) ct.done = 1;
      else {
        finalize = true;
        fin_div = div;
        fin_term = term;
      }
    } else {
      ct.step = step + 1;
    }
    if (finalize) {
      AEHMC_FRESH_TT();
      const int dir = ct.dir, oth = 1 - dir;
      d_l = 0.0;
      d_r = 0.0;
#pragma unroll
      for (int b0 = 0; b0 < NB; b0++) {
        double pov[BR], psv[BR], imv[BR];
#pragma unroll
        for (int u = 0; u < BR; u++) {
          const int r = b0 * BR + u;
          pov[u] = AT(pick2(a.end_p, oth), r);
          psv[u] = AT(a.psum, r);
          imv[u] = IMOF(r);
        }
#pragma unroll
        for (int u = 0; u < BR; u++) {
          const int r = b0 * BR + u;
          AEHMC_SLOT(r, {
            const double pc = p[r], po = pov[u];
            const double vc = imv[u] * pc, vo = imv[u] * po;
            const double s = psv[u] + pb[r];
            const double pl = dir ? po : pc, pr = dir ? pc : po;
            const double vl = dir ? vo : vc, vr = dir ? vc : vo;
            const double rho = s - (pr + pl) / 2;
            d_l += on ? vl * rho : 0.0;
            d_r += on ? vr * rho : 0.0;
            if (on) {
              AT(a.psum, r) = s;
              AT(pick2(a.end_q, dir), r) = QGET(r);
              AT(pick2(a.end_p, dir), r) = pc;
              AT(pick2(a.end_g, dir), r) = GGET(r);
            }
          });
        }
        if (R > 8) __builtin_amdgcn_sched_barrier(0);
      }
      double e0 = 0.0, e1 = 0.0;
      sum4(d_l, d_r, e0, e1);
      const bool turning = (d_l <= 0) | (d_r <= 0);
      put2(ct.U_end, dir, ct.U_cur);
      ct.acc_prob = exp(ct.sub_slpa) / (double)ct.length;
      double pbias = exp(ct.sub_w - ct.prop_w);
      if (pbias > 1.0) pbias = 1.0;
      if (pbias < 0.0) pbias = 0.0;
      const int acc_b = rng_bernoulli(rng.g[3], pbias);
      if (fin_div || fin_term) {
        ct.prop_slpa = np_logaddexp(ct.sub_slpa, ct.prop_slpa);
      } else {
        ct.prop_w = np_logaddexp(ct.prop_w, ct.sub_w);
        ct.prop_slpa = np_logaddexp(ct.prop_slpa, ct.sub_slpa);
        if (acc_b) {
          ct.prop_slot ^= 1;
          ct.prop_E = ct.sub_E;
        }
      }
      ct.ndoubl = ct.j + 1;
      ct.out_div = fin_div;
      ct.out_turn = turning;
      const bool end_transition = fin_div || turning || fin_term || (ct.j + 1 == a.max_exp);
      if (end_transition) {
        const int s = ct.prop_slot;  // outputs (the phantom scan below cannot change them)
#pragma unroll
        for (int r = 0; r < R; r++) {
          AEHMC_SLOT(r, {
            if (on) {
              AT(a.q, r) = AT(pick2(a.slot_q, s), r);
              AT(a.g, r) = AT(pick2(a.slot_g, s), r);
              if (a.out.momentum) AT(a.out.momentum, r) = AT(pick2(a.slot_p, s), r);
            }
          });
        }
        if (t == 0) {
          a.U[c] = pick2(ct.U_slot, s);
          a.out.acceptance_probability[c] = ct.acc_prob;
          if (a.out.num_doublings) a.out.num_doublings[c] = ct.ndoubl;
          if (a.out.is_turning) a.out.is_turning[c] = ct.out_turn;
          a.out.is_diverging[c] = ct.out_div;
        }
        if (step == 0 && fin_div) {  // trajectory.py:336: the scan still runs (phantom)
          ct.phantom = 1;
          ct.step = 1;
        } else {
          ct.done = 1;
        }
      } else {
        ct.j += 1;
        const int go_right = rng_bernoulli(rng.g[1], 0.5);
        ct.dir = go_right;
        ct.step = 0;
        if (go_right != dir) {  // continue from the other end
#pragma unroll
          for (int r = 0; r < R; r++) {
            AEHMC_SLOT(r, {
              const double qv = AT(pick2(a.end_q, go_right), r);
              const double gv = DG ? AT(pick2(a.end_g, go_right), r) : qv;
              p[r] = AT(pick2(a.end_p, go_right), r);
              QSET(r, qv);
              GSET(r, gv);
            });
          }
          ct.U_cur = pick2(ct.U_end, go_right);
        }
      }
    }
  }
  if (t == 0) {
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = ct.nleap;
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, rng.g[1]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 2) * 4, rng.g[2]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 3) * 4, rng.g[3]);
  }
#undef AEHMC_FRESH_TT
#undef AEHMC_SLOT
#undef EI
#undef VALID
#undef AT
#undef QGET
#undef QSET
#undef GGET
#undef GSET
#undef IMOF
}

inline bool nuts_wide_supported(int tkind, int met_ndim, long long D) {
  // (D = 1e4 with a diagonal-Gaussian target needs 160 000 of the CU's 163 840 bytes of LDS for q
  //  and dU/dq; the kernel's small static arrays leave room for D up to 10176)
  return (tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN) &&
         met_ndim < 2 && D > 512 && D <= 10176;
}

template <int T, int R, bool QGL>
inline hipError_t launch_nuts_wide_tr(const EngineArgs &a, hipStream_t st) {
  const bool dg = a.tkind == AEHMC_T_DIAG_GAUSSIAN;
  const size_t dyn = QGL ? (size_t)(dg ? 2 : 1) * a.D * sizeof(double) : 0;
  const dim3 grid((unsigned)a.C), block(T);
#define AEHMC_WIDE_LAUNCH(TKV)                                                                          \
  do {                                                                                                  \
    if (QGL) {                                                                                          \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_wide<T, R, QGL, TKV>),  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);        \
      if (e != hipSuccess) return e;                                                                    \
    }                                                                                                   \
    hipLaunchKernelGGL((k_nuts_wide<T, R, QGL, TKV>), grid, block, dyn, st, a);                         \
  } while (0)
  switch (a.tkind) {
    case AEHMC_T_STD_NORMAL: AEHMC_WIDE_LAUNCH(AEHMC_T_STD_NORMAL); break;
    case AEHMC_T_ISO_GAUSSIAN: AEHMC_WIDE_LAUNCH(AEHMC_T_ISO_GAUSSIAN); break;
    default: AEHMC_WIDE_LAUNCH(AEHMC_T_DIAG_GAUSSIAN);
  }
#undef AEHMC_WIDE_LAUNCH
  return hipGetLastError();
}
// the momentum of site #1 must already be in a.zbuf (k_draw_momentum)
inline hipError_t launch_nuts_wide(const EngineArgs &a, hipStream_t st) {
  const long long D = a.D;
  if (D <= 1024) return launch_nuts_wide_tr<256, 4, false>(a, st);
  if (D <= 2048) return launch_nuts_wide_tr<256, 8, false>(a, st);
  if (D <= 4096) return launch_nuts_wide_tr<512, 8, false>(a, st);
  if (D <= 8192) return launch_nuts_wide_tr<512, 16, true>(a, st);
  return launch_nuts_wide_tr<512, 20, true>(a, st);
}

}  // namespace aehmc
