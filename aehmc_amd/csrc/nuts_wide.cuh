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
//  * Bytes: the initial state is never copied (the caller's q / dU/dq and the drawn momentum
//    alias it as proposal and as trajectory ends until something else takes their place); a
//    trajectory end is stored only when the next expansion turns to the other side; checkpoint
//    pairs that only the following step reads (every other one) are not stored; with
//    dU/dq == q (standard / isotropic normal) no gradient array is touched.
// HBM then sees per leapfrog: a checkpoint pair every fourth step (4 D bytes on average),
// ~1/2 checkpoint pair read (8 D), the proposal copy on accept and psum / the parked end at
// expansion boundaries -- ~22 D bytes instead of the 88 D of a streaming step.
//
// Arithmetic per element is that of engine.cuh's lock-step path; sums are accumulated per
// thread in ascending element order, then wave (DPP) and cross-wave in a fixed order, so
// results agree with the oracle to rounding (tested at 1e-9), independent of the launch.
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235,
// proposals.py:19-174, integrators.py:54-73, metrics.py:44-104.
#pragma once
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime itself) */
#include <hip/hip_runtime.h>
#endif

#include "engine.cuh"
#include "linreg_rows.cuh"  // dpp_xor1 / dpp_xor2

namespace aehmc {

struct WideTagTrue { static constexpr bool value = true; };    // (compile-time switches of the pass below;
struct WideTagFalse { static constexpr bool value = false; };  //  no <type_traits> under hipRTC)

// Developer instrumentation (make timing): shader-clock cycles per phase of the per-leapfrog
// loop, accumulated by every thread, written by thread 0 to a.linreg_part[c * 8 + phase]
// (unused workspace on this path).  Compiled out of the product library.
#ifdef AEHMC_WIDE_TIMING
#define AEHMC_TICK(k)                                         \
  do {                                                        \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
    tacc[k] += now_ - tlast;                                  \
    tlast = now_;                                             \
  } while (0)
#else
#define AEHMC_TICK(k) do { } while (0)
#endif

template <int T, int R, bool QGL, int TK>
__global__ __launch_bounds__(T) void k_nuts_wide(EngineArgs a) {
  constexpr int NW = T / 64;
  // TK == AEHMC_T_CUSTOM exists only in the run-time compiled copy (aehmc_set_custom_target): the user's aehmc_custom_elem,
  // dU/dq kept beside q as for the diagonal Gaussian, no parameters of the engine's own to stream
  constexpr bool CU = TK == AEHMC_T_CUSTOM;
  constexpr bool DG = TK == AEHMC_T_DIAG_GAUSSIAN || CU;  // otherwise dU/dq == q: no separate copy
  constexpr bool ISO = TK == AEHMC_T_ISO_GAUSSIAN;
  constexpr bool PAR_REG = R <= 8;          // per-element parameters live in VGPRs
  constexpr bool IM_LDS = !PAR_REG && !DG;  // imm in the LDS half that dU/dq does not need
  constexpr bool STREAM = !PAR_REG && DG;   // parameters streamed from L2 every pass
  constexpr int BR = STREAM ? 2 : ((R % 4 == 0) ? 4 : R);  // elements per batch (streamed parameters: two, so that
  //                                                           the batch in use and the one on its way stay at 32 registers)
  constexpr int NB = R / BR;
  static_assert(R % BR == 0, "R must be a multiple of the batch size");
  static_assert(QGL || PAR_REG, "more than 8 elements per thread: q lives in LDS");
  __shared__ double red[2][4 * NW];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  const unsigned D = (unsigned)a.D;
  // LDS arrays have D + 1 entries: entry D is the shared dummy of the slots past D (always 0 in q
  // and dU/dq, 1 in imm)
  double *const sq = dyn_lds, *const sg = dyn_lds + (QGL && DG ? D + 1 : 0);
  double *const sim = dyn_lds + (IM_LDS ? D + 1 : 0);
  int flip = 0;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long long c = blockIdx.x;
  const bool im_scalar = a.met_ndim == 0;
  const double *const imrow = a.imm + (size_t)c * a.imm_cs;
  // Thread t owns slots r = 0 .. nslots-1 = elements t + T r.  The engine's work arrays have
  // rows padded to a.ldw (a multiple of T) here, and a slot past D carries q = p = 0 for the whole
  // transition (0 is a fixed point of the leapfrog for every target below once its parameters
  // are neutral): its terms in all sums are exactly 0 and its stores land in the padding, so
  // the per-leapfrog code has no predicates.  Only the caller's unpadded arrays (q, dU/dq, the
  // momentum output) and the parameter vectors are accessed clamped / masked.
  const int nslots = (int)((D + T - 1) / T);
  const size_t rowW = (size_t)c * a.ldw, rowU = (size_t)c * D;
  // `tt` is t behind an opaque barrier that is renewed in every loop iteration: element
  // addresses derived from it cannot be hoisted out of the loops (the compiler would otherwise
  // precompute a 64-bit address per element and array -- hundreds of VGPRs -- and spill them)
  int tt = t;
#define AEHMC_FRESH_TT() asm volatile("" : "+v"(tt))
#define EW(r) ((unsigned)(tt + T * (r)))                      /* element index, work arrays */
#define EC(r) (EW(r) < D ? EW(r) : D - 1)                     /* clamped: caller arrays, parameters */
#define EL(r) (EW(r) < D ? EW(r) : D)                         /* LDS index (dummy entry D) */
#define ON(r) (EW(r) < D)
#define WA(ptr, r) ((ptr) + rowW)[EW(r)]
#define UA(ptr, r) ((ptr) + rowU)[EC(r)]

  // team sum of four values; every thread returns the same bits (SGPRs).  Inside a wave the four
  // values are reduced together: two "transpose" stages (lane bit b keeps one half of the values and
  // receives the partner's other half) leave one register per lane, four butterfly stages finish it
  // -- 7 cross-lane additions instead of 24; lane l then holds the wave total of value l & 3.
  auto sum4 = [&](double &x0, double &x1, double &x2, double &x3) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double r0 = (b0 ? x1 : x0) + dpp_xor1(b0 ? x0 : x1);
    const double r1 = (b0 ? x3 : x2) + dpp_xor1(b0 ? x2 : x3);
    double x = (b1 ? r1 : r0) + dpp_xor2(b1 ? r0 : r1);
    x += __shfl_xor(x, 4);
    x += __shfl_xor(x, 8);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    double *buf = red[flip];
    flip ^= 1;  // double-buffered: the next reduction writes the other buffer
    if (lane < 4) buf[4 * wave + lane] = x;
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

  // per-element parameters; a slot past D gets neutral ones (imm 1, mu 0, sigma 1, log sigma 0)
  struct Par {
    double im, mu, sd, ls;
  };
  auto par_load = [&](int r) {
    Par x;
    const unsigned i = EC(r);
    const bool on = ON(r);
    const double im = imrow[im_scalar ? 0u : i];
    x.im = on ? im : 1.0;
    if (DG && !CU) {
      const double mu = a.mu[i], sd = a.sigma[i], ls = a.log_sigma[i];
      x.mu = on ? mu : 0.0;
      x.sd = on ? sd : 1.0;
      x.ls = on ? ls : 0.0;
    } else {
      x.mu = 0.0;
      x.sd = 1.0;
      x.ls = 0.0;
    }
    return x;
  };

  double q[QGL ? 1 : R], g[(QGL || !DG) ? 1 : R], p[R], pb[R];
  Par preg[PAR_REG ? R : 1];
#define QGET(r) (QGL ? sq[EL(r)] : q[QGL ? 0 : (r)])
#define QSET(r, v) do { if (QGL) sq[EL(r)] = (v); else q[QGL ? 0 : (r)] = (v); } while (0)
#define GGET(r) (!DG ? QGET(r) : (QGL ? sg[EL(r)] : g[(QGL || !DG) ? 0 : (r)]))
#define GSET(r, v) do { if (DG) { if (QGL) sg[EL(r)] = (v); else g[(QGL || !DG) ? 0 : (r)] = (v); } } while (0)
// imm of slot r outside the streamed pass (registers / LDS / L2)
#define IMOF(r) (PAR_REG ? preg[PAR_REG ? (r) : 0].im : IM_LDS ? sim[EL(r)] : (ON(r) ? imrow[im_scalar ? 0u : EC(r)] : 1.0))

  // Where the states that are NOT on chip live.  The initial state (q0 = a.q, p0 = a.zbuf,
  // dU/dq0 = a.g; the caller's arrays stay intact until the transition's outputs are written) is
  // never copied: "buffer 2" of the proposal slots and an `init` flag per trajectory end alias it.
  // The end of the side that is being integrated lives in registers / LDS and is stored only when
  // the next expansion turns to the other side.  With dU/dq == q (!DG) no dU/dq array is touched.
  int prop_buf = 2;                 // proposal: slot 0 / 1, or 2 = the initial state
  bool end_init0 = true, end_init1 = true, psum_init = true;

  // ---- load the chain: q, dU/dq, momentum (drawn by k_draw_momentum into zbuf, rows padded
  //      with zeros), parameters; nuts.py:113-125 ----------------------------------------------
  double kd = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    p[r] = pb[r] = 0.0;
    if (!QGL) q[QGL ? 0 : r] = 0.0;
    if (!QGL && DG) g[(QGL || !DG) ? 0 : r] = 0.0;
    if (PAR_REG) preg[PAR_REG ? r : 0] = Par{1.0, 0.0, 1.0, 0.0};
    if (r < nslots) {
      const bool on = ON(r);
      const double qv = on ? UA(a.q, r) : 0.0, gv = DG ? (on ? UA(a.g, r) : 0.0) : qv, pv = WA(a.zbuf, r);
      if (PAR_REG) preg[PAR_REG ? r : 0] = par_load(r);
      const double im = on ? imrow[im_scalar ? 0u : EC(r)] : 1.0;
      if (IM_LDS) sim[EL(r)] = im;
      p[r] = pv;
      QSET(r, qv);
      GSET(r, gv);
      kd += (im * pv) * pv;
    }
  }
  ChainRng rng = rng_load(a, c);
  ChainCtl ct = {};
  sum4(kd, z1, z2, z3);
  const double U0 = a.U[c];
  {
    ct.H0 = U0 + 0.5 * kd;
    ct.prop_w = 0.0;
    ct.prop_slpa = -INFINITY;
    ct.U_cur = ct.U_end[0] = ct.U_end[1] = ct.U_slot[0] = ct.U_slot[1] = U0;
    ct.dir = rng_bernoulli(rng.g[1], 0.5);  // trajectory.py:516
  }
  const double eps = a.eps_c ? a.eps_c[c] : a.eps;
  int ck_last = -1;  // checkpoint index whose pair the registers held before this step (-1: none)
  Par cur[BR], nxt[BR];  // streamed parameters: the batch in use and the next one on its way

#ifdef AEHMC_WIDE_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = (long long)__builtin_amdgcn_s_memtime();
#endif
  while (!ct.done) {
    AEHMC_FRESH_TT();
    AEHMC_TICK(7);
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size, aa = 1 * step_size;
    const int step = ct.step;
    if (!ct.phantom) ct.nleap += 1;
    const TreeIdx ti = tree_step_indices(step, ct.tmin, ct.tmax);  // termination.py:109-113 (stale at step 0), 192-235
    const int tmin = ti.tmin, tmax = ti.tmax;
    const bool even = (step & 1) == 0;
    const bool check = step >= 1 && tmax >= tmin;
    // level tmax of the check is the pair the previous step stored: p and the momentum sum as
    // they are in registers before this step's update
    const bool fwd = check && ck_last == tmax;
    // termination.py:115-124 stores a checkpoint pair at every even step.  Only the pairs that
    // open a sub-tree of 4 or more steps are ever read back from memory (at step + 2^k - 1,
    // k >= 2, which needs step = 0 mod 4): the pair of a step = 2 mod 4 is consumed by the next
    // step alone -- from registers, above -- and overwritten before any other read, and the
    // step-0 pair written to a stale index >= 1 is overwritten by that index's own sub-tree before
    // it is read (only index 0 is ever read without a store of this sub-trajectory before it).
    const bool ck_store = (step & 3) == 0 && (step > 0 || tmax == 0);
    double *const ckp = a.ckp + ((size_t)tmax * a.C + c) * a.ldw;
    double *const cks = a.cks + ((size_t)tmax * a.C + c) * a.ldw;

    // ---- one pass: leapfrog + kinetic energy + momentum sum + checkpoint + first U-turn level ----
    double usum = 0.0, d_l = 0.0, d_r = 0.0;
    kd = 0.0;
    if (step == 0) {  // the momentum sum of the sub-trajectory restarts: 0 + p' == p'
#pragma unroll
      for (int r = 0; r < R; r++) pb[r] = 0.0;
    }
    auto pass = [&](auto fwd_tag) {
      constexpr bool FWD = decltype(fwd_tag)::value;
      if (STREAM) {
#pragma unroll
        for (int u = 0; u < BR; u++) cur[u] = par_load(u);
      }
#pragma unroll
      for (int b0 = 0; b0 < NB; b0++) {
        if (STREAM && b0 + 1 < NB) {
#pragma unroll
          for (int u = 0; u < BR; u++) nxt[u] = par_load((b0 + 1) * BR + u);
        }
#pragma unroll
        for (int u = 0; u < BR; u++) {
          const int r = b0 * BR + u;
          if (r < nslots) {
            Par x = PAR_REG ? preg[PAR_REG ? r : 0] : cur[u];
            if (IM_LDS) x.im = sim[EL(r)];
            const double p_old = p[r], pb_old = pb[r];
            double pp = p_old - b * GGET(r);               // integrators.py:59-60
            const double qq = QGET(r) + aa * (x.im * pp);  // integrators.py:62-64
            double uu, gg;
            if (TK == AEHMC_T_STD_NORMAL) {
              uu = 0.5 * (qq * qq) + AEHMC_LOG_SQRT_2PI;
              gg = qq;
            } else if (ISO) {
              uu = qq * qq;
              gg = qq;
            } else if (CU) {
#ifdef AEHMC_CUSTOM_TARGET
              aehmc_custom_elem(qq, (long long)EC(r), a.cparams, uu, gg);
#else
              uu = gg = 0.0;
#endif
              gg = ON(r) ? gg : 0.0;  // (a slot past D stays at the fixed point q = p = 0)
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
            const double s = pb_old + pp;  // trajectory.py:243 (the sum restarts from 0 at step 0, :278)
            pb[r] = s;
            usum += (ISO || ON(r)) ? uu : 0.0;  // (a slot past D adds the target's constant only)
            kd += v * pp;
            if (FWD) {  // termination.py:160-173 for level idx_max, from registers
              const double pl = p_old, vl = x.im * pl;
              const double sub = s - pb_old + pl;
              const double rho = sub - (pp + pl) / 2;
              d_l += vl * rho;
              d_r += v * rho;
            }
          }
        }
        if (STREAM && b0 + 1 < NB) {
#pragma unroll
          for (int u = 0; u < BR; u++) cur[u] = nxt[u];
        }
        if (R > 8) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (fwd) pass(WideTagTrue{});
    else pass(WideTagFalse{});
    if (ck_store) {  // every fourth step: the checkpoint pair, straight from the registers
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (r < nslots) {
          ckp[EW(r)] = p[r];
          cks[EW(r)] = pb[r];
        }
      }
    }
    if (even) ck_last = tmax;
    AEHMC_TICK(0);  // pass
    // fetched ahead, behind the reduction and the per-chain scalar work below: the first checkpoint
    // pair this step's U-turn check needs from memory (level tmax when it is not in registers, else
    // level tmax - 1).  (The streamed variants used to fetch the first parameter batch of the NEXT pass
    // here as well: its 16 registers, live across the scalar code next to the 4 R of the checkpoint
    // pair, made them spill 96 / 224 B per lane; the batch is now loaded at the head of the pass.)
    const int pre_idx = fwd ? tmax - 1 : tmax;
    const bool pre_on = check && pre_idx >= tmin;
    double pre_kp[R], pre_ks[R];
    if (pre_on) {
      const double *kp = a.ckp + ((size_t)pre_idx * a.C + c) * a.ldw;
      const double *ks = a.cks + ((size_t)pre_idx * a.C + c) * a.ldw;
#pragma unroll
      for (int r = 0; r < R; r++) {
        pre_kp[r] = r < nslots ? kp[EW(r)] : 0.0;
        pre_ks[r] = r < nslots ? ks[EW(r)] : 0.0;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) pre_kp[r] = pre_ks[r] = 0.0;
    }
    AEHMC_TICK(1);  // prefetch issue
    sum4(usum, kd, d_l, d_r);
    AEHMC_TICK(2);  // reduction + barrier
    ct.U_cur = ISO ? 0.5 * usum : usum;
    ct.tmin = tmin;
    ct.tmax = tmax;

    // ---- dynamic_integration body (trajectory.py:195-305), per-chain scalars ------------------
    // proposals.py:19-62, 72-102, 141-144 (nuts_tree.cuh)
    const TreePoint np = tree_new_point(ct.H0, ct.U_cur, kd, a.thr);
    const bool div = np.div;
    bool term = false;
    const bool take = tree_sample_step<true>(ct, step, np, lane, [&](double pr) { return rng_bernoulli(rng.g[2], pr); });
    if (step >= 1) {
      AEHMC_TICK(3);  // per-chain scalars
      if (check) {  // termination.py:133-187
        int idx = tmax;
        bool crit = false;
        for (;;) {
          if (!(fwd && idx == tmax)) {  // a level that is not in registers: its checkpoint pair from memory
            AEHMC_FRESH_TT();
            if (idx != pre_idx) {  // not fetched ahead (third and later levels: 1 step in 8)
              const double *kp = a.ckp + ((size_t)idx * a.C + c) * a.ldw;
              const double *ks = a.cks + ((size_t)idx * a.C + c) * a.ldw;
#pragma unroll
              for (int r = 0; r < R; r++) {
                pre_kp[r] = r < nslots ? kp[EW(r)] : 0.0;
                pre_ks[r] = r < nslots ? ks[EW(r)] : 0.0;
              }
            }
            d_l = 0.0;
            d_r = 0.0;
#pragma unroll
            for (int r = 0; r < R; r++) {
              if (r < nslots) {
                const double im = IMOF(r);
                const double pl = pre_kp[r], pr = p[r];
                const double vl = im * pl, vr = im * pr;
                const double sub = pb[r] - pre_ks[r] + pl;
                const double rho = sub - (pr + pl) / 2;
                d_l += vl * rho;
                d_r += vr * rho;
              }
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
    AEHMC_TICK(4);  // U-turn levels from memory
    const int sub_buf = prop_buf == 0 ? 1 : 0;  // slot of this sub-trajectory's proposal
    if (take) {  // sub-trajectory proposal <- moving end (copy on accept)
      AEHMC_FRESH_TT();
      double *const dq = pick2(a.slot_q, sub_buf), *const dp = pick2(a.slot_p, sub_buf),
                   *const dg = pick2(a.slot_g, sub_buf);
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (r < nslots) {
          WA(dq, r) = QGET(r);
          WA(dp, r) = p[r];
          if (DG) WA(dg, r) = GGET(r);
        }
      }
      put2(ct.U_slot, sub_buf, ct.U_cur);
    }

    AEHMC_TICK(5);  // proposal copy
    // ---- sub-trajectory / expansion control (trajectory.py:336, 537-608) ----------------------
    const TreeControl tc = tree_step_control(ct, step, div, term);
    const bool fin_div = tc.fin_div, fin_term = tc.fin_term;
    if (tc.finalize) {
      AEHMC_FRESH_TT();
      const int dir = ct.dir, oth = 1 - dir;
      const bool oth_init = oth ? end_init1 : end_init0;
      const double *const po_src = oth_init ? a.zbuf : pick2(a.end_p, oth);
      const double *const ps_src = psum_init ? a.zbuf : a.psum;
      double pov[R], psv[R];  // every load of the sweep is issued before the first use
#pragma unroll
      for (int r = 0; r < R; r++) {
        pov[r] = r < nslots ? WA(po_src, r) : 0.0;
        psv[r] = r < nslots ? WA(ps_src, r) : 0.0;
      }
      d_l = 0.0;
      d_r = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (r < nslots) {
          const double im = IMOF(r);
          const double pc = p[r], po = pov[r];
          const double vc = im * pc, vo = im * po;
          const double s = psv[r] + pb[r];
          const double pl = dir ? po : pc, pr = dir ? pc : po;
          const double vl = dir ? vo : vc, vr = dir ? vc : vo;
          const double rho = s - (pr + pl) / 2;
          d_l += vl * rho;
          d_r += vr * rho;
          WA(a.psum, r) = s;
        }
      }
      psum_init = false;
      double e0 = 0.0, e1 = 0.0;
      sum4(d_l, d_r, e0, e1);
      const bool turning = (d_l <= 0) | (d_r <= 0);
      put2(ct.U_end, dir, ct.U_cur);
      if (tree_merge_expansion<true>(ct, fin_div, fin_term, lane, [&](double pr) { return rng_bernoulli(rng.g[3], pr); }))
        prop_buf = sub_buf;
      const bool end_transition = tree_expansion_outcome(ct, fin_div, fin_term, turning, a.max_exp);
      if (end_transition) {
        // outputs (the phantom scan below cannot change them); a proposal that is still the initial
        // state leaves q, dU/dq and U as they are
        const double *const oq = pick2(a.slot_q, prop_buf & 1), *const og = pick2(a.slot_g, prop_buf & 1);
        const double *const op = prop_buf == 2 ? a.zbuf : pick2(a.slot_p, prop_buf);
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (r < nslots) {
            const double pv = WA(op, r);
            if (ON(r)) {
              if (prop_buf != 2) {
                const double qv = WA(oq, r);
                UA(a.q, r) = qv;
                UA(a.g, r) = DG ? WA(og, r) : qv;
              }
              if (a.out.momentum) UA(a.out.momentum, r) = pv;
            }
          }
        }
        if (t == 0) {
          if (prop_buf != 2) a.U[c] = pick2(ct.U_slot, prop_buf);
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
        if (go_right != dir) {  // continue from the other end: park this end, fetch that one
          const bool src_init = go_right ? end_init1 : end_init0;
          double nq[R], np_[R], ng[DG ? R : 1];
          if (src_init) {  // the caller's (unpadded) arrays
#pragma unroll
            for (int r = 0; r < R; r++) {
              const bool on = r < nslots && ON(r);
              nq[r] = on ? UA(a.q, r) : 0.0;
              if (DG) ng[DG ? r : 0] = on ? UA(a.g, r) : 0.0;
              np_[r] = r < nslots ? WA(a.zbuf, r) : 0.0;
            }
          } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
              nq[r] = r < nslots ? WA(pick2(a.end_q, go_right), r) : 0.0;
              if (DG) ng[DG ? r : 0] = r < nslots ? WA(pick2(a.end_g, go_right), r) : 0.0;
              np_[r] = r < nslots ? WA(pick2(a.end_p, go_right), r) : 0.0;
            }
          }
#pragma unroll
          for (int r = 0; r < R; r++) {
            if (r < nslots) {
              WA(pick2(a.end_q, dir), r) = QGET(r);
              WA(pick2(a.end_p, dir), r) = p[r];
              if (DG) WA(pick2(a.end_g, dir), r) = GGET(r);
              p[r] = np_[r];
              QSET(r, nq[r]);
              GSET(r, DG ? ng[DG ? r : 0] : nq[r]);
            }
          }
          if (dir) end_init1 = false;
          else end_init0 = false;
          ct.U_cur = pick2(ct.U_end, go_right);
        }
      }
      AEHMC_TICK(6);  // expansion boundary
    }
  }
#ifdef AEHMC_WIDE_TIMING
  if (t == 0)
    for (int k = 0; k < 8; k++) a.linreg_part[c * 8 + k] = (double)tacc[k];
#endif
  if (t == 0) {
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = ct.nleap;
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, rng.g[1]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 2) * 4, rng.g[2]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 3) * 4, rng.g[3]);
  }
#undef AEHMC_FRESH_TT
#undef EW
#undef EC
#undef EL
#undef ON
#undef WA
#undef UA
#undef QGET
#undef QSET
#undef GGET
#undef GSET
#undef IMOF
}

#ifndef __HIPCC_RTC__  // (host side)
inline bool nuts_wide_supported(int tkind, int met_ndim, long long D) {
  // (D = 1e4 with a diagonal-Gaussian target needs 160 000 of the CU's 163 840 bytes of LDS for q
  //  and dU/dq; the kernel's small static arrays leave room for D up to 10176)
  return (tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN) &&
         met_ndim < 2 && D > 512 && D <= 10176;
}

template <int T, int R, bool QGL>
inline hipError_t launch_nuts_wide_tr(const EngineArgs &a, hipStream_t st) {
  // q and (diagonal target) dU/dq or (otherwise, R > 8) imm: two arrays of D + 1 doubles
  const size_t dyn = QGL ? (size_t)2 * (a.D + 1) * sizeof(double) : 0;
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
// row stride of the engine's work arrays on this path (a multiple of every team size)
inline long long nuts_wide_ld(long long D) { return (D + 511) / 512 * 512; }
// the momentum of site #1 must already be in a.zbuf (k_draw_momentum, rows of a.ldw, zero padded)
inline hipError_t launch_nuts_wide(const EngineArgs &a, hipStream_t st) {
  const long long D = a.D;
  if (D <= 1024) return launch_nuts_wide_tr<256, 4, false>(a, st);
  if (D <= 2048) return launch_nuts_wide_tr<256, 8, false>(a, st);
  if (D <= 4096) return launch_nuts_wide_tr<512, 8, false>(a, st);
  if (D <= 8192) return launch_nuts_wide_tr<512, 16, true>(a, st);
  return launch_nuts_wide_tr<512, 20, true>(a, st);
}
#endif  // __HIPCC_RTC__

}  // namespace aehmc
