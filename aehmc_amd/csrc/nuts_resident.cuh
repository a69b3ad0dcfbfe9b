// Register-resident NUTS transition (gfx950): the whole nuts.new_kernel(...)(state, eps, imm)
// call in ONE launch, the chain's moving state living on chip for the entire tree.
//
// A *team* of T threads owns one chain (D <= 512; larger chains: nuts_wide.cuh):
//   T = 64            one wavefront per chain (the north star's layout), 128 < D <= 512;
//   T = 1 ... 32      sub-wavefront teams for small D: 64/T chains share a wavefront, so the
//                     per-chain scalar work (RNG, exp/log of the proposal weights, tree
//                     indices) -- ~3000 fp64 instructions per leapfrog that a full wave
//                     would execute 64x redundantly -- is executed once per T lanes; the
//                     chains of a wave diverge like ordinary SIMT threads.
// Thread t keeps elements t, t+T, ... of
// q, p, dU/dq and of the sub-trajectory momentum sum in VGPRs; energies and U-turn dot
// products are DPP wavefront reductions.  HBM/L2 sees only the U-turn checkpoints (written every other step,
// ~1 pair read per step), the trajectory ends at expansion boundaries and the proposal on
// accept -- instead of the 88*D bytes per leapfrog of a streaming implementation.
//
// MULTI instantiations run m.T consecutive transitions per launch (kernel.sample(T)) and, with m.adapt,
// the window-adaptation update after each of them (window_adaptation.run in one launch): per-transition
// records go to m.samples / m.acc_hist / m.div_hist, the generator states stay in registers in between.
//
// Diagonal / scalar metric (shared or per chain), coordinate-wise targets.  Arithmetic and
// its order are those of the lock-step path in engine.cuh (for T = 64 bit for bit, tested);
// DENSE instantiations (T = 64, one element per lane, D <= 64: "small dense problems", the classic full-mass-matrix
// use) add a dense inverse mass matrix (shared, or one per chain) and / or the dense-precision target: the D x D
// products of metrics.py:66-71 run inside the wavefront (wave_matvec_reg: matrix transposed in LDS -- or, per
// chain, in a workspace written once per transition -- operand broadcast from registers), the velocity imm p
// lives in a register beside p and is checkpointed with it as the lock-step path does (ckv / end_v);
// reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235,
// proposals.py:19-174, integrators.py:54-73, metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"
#include "linreg_rows.cuh"  // wave_sum8 and friends

namespace aehmc {

template <int T>
struct Team {
  static_assert(T <= 64, "teams are at most one wavefront");
  static constexpr bool SUB = (T < 64);    // several chains per wavefront
  static constexpr bool WAVE = (T == 64);  // one wavefront per chain
  static constexpr int BLOCK = 256;
};
// DENSE bits of k_nuts_resident
constexpr int RES_DENSE_METRIC = 1, RES_DENSE_TARGET = 2, RES_DENSE_PER_CHAIN = 4;
constexpr int RES_DENSE_JOINT = 8;  // a joint (non-separable) user-defined target: the leapfrog of the DENSE path, no matrix of its own
constexpr int RES_DENSE_BLOCK = 512;  // eight chains share the matrices in LDS (up to 96 KB at D = 64)

// butterfly sum over the T (< 64) consecutive lanes of a sub-wavefront team
template <int T>
__device__ __forceinline__ double subwave_sum(double x) {
  if (T >= 2) x = dpp_add<0xB1>(x);
  if (T >= 4) x = dpp_add<0x4E>(x);
  if (T >= 8) x = dpp_add<0x141>(x);
  if (T >= 16) x = dpp_add<0x140>(x);
  if (T >= 32) x += __shfl_xor(x, 16);
  return x;
}

// sum of two values over the team; every thread of the team returns the same bits.
// `single`: only thread 0 of the team holds a term (D == 1 on a whole wavefront -- the README example's single
// chain): the butterfly would add 63 zeros to it, i.e. return lane 0's value + 0.0; reading that lane directly skips
// ~50 dependent cross-lane instructions per call on the latency path of a lone chain (same bits: x + 0.0 also
// turns a -0.0 into the +0.0 the butterfly produces).
template <int T>
__device__ __forceinline__ void team_sum2(double &x, double &y, bool single = false) {
  if (Team<T>::SUB) {
    x = subwave_sum<T>(x);
    y = subwave_sum<T>(y);
    return;
  }
  if (single) {
    x = read_lane_f64(x, 0) + 0.0;
    y = read_lane_f64(y, 0) + 0.0;
    return;
  }
  x = wave_sum(x);
  y = wave_sum(y);
}

// MULTI: the launch runs m.T > 1 transitions (the RNG state, the leapfrog total ... stay live across
// the tree loop, which costs ~35 VGPRs and a wavefront per SIMD: single transitions keep their own
// instantiation).
// CKL: the U-turn checkpoints of the chain (momentum and momentum sum per tree level, termination.py:12-16) live
// in LDS instead of global memory -- [wave][level][2][64] doubles, element i written and read by lane i only.  For a
// few chains (one wavefront per SIMD or less: the README example's single chain) the checkpoint reads are dependent
// L2 round trips on the serial path of every leapfrog; with many chains the 40 KB per workgroup would cost occupancy.
// Wavefronts per SIMD the register allocator has to leave room for.  A MULTI launch of one wavefront per chain keeps
// ~80 VGPRs of chain state live through the tree loop; left alone, the scheduler gives up on 128 registers once the
// inlined log1p / exp bodies of the proposal weights push past them and settles at 150-170 (3 waves per SIMD: 4096 chains
// then run as a full round plus a one-third-full one).  Held to 4 waves it fits 127-128 registers and what it spills
// (10-54 dwords: row base addresses and adaptation scalars) is stored once per launch and reloaded once per TRANSITION,
// outside the tree loop (profiles/r4/INDEX.md has the per-loop-depth count of scratch instructions).
#ifndef AEHMC_RES_MIN_WAVES
#define AEHMC_RES_MIN_WAVES 1
#endif
constexpr int res_min_waves(int T, int R, bool MULTI) { return (AEHMC_RES_MIN_WAVES && T == 64 && MULTI && R <= 4) ? 4 : 1; }
// (the dense instantiations run 512-thread workgroups: two wavefronts per SIMD have to fit.  Said explicitly since round 6:
//  with amdgpu_waves_per_eu(1) the compiler took the freedom to use 328 registers for a user density with long unrolled
//  inner loops -- a code object that cannot be launched, HSA_STATUS_ERROR_INVALID_ISA)
// (compiled against a user's joint density -- AEHMC_JOINT_TARGET -- three: engine.cuh wg_min_waves has the measurement)
constexpr int res_min_waves_dense(int T, int R, bool MULTI, int DENSE) {
#ifdef AEHMC_JOINT_TARGET
  constexpr int need = 3;
#else
  constexpr int need = 2;
#endif
  return DENSE ? (res_min_waves(T, R, MULTI) > need ? res_min_waves(T, R, MULTI) : need) : res_min_waves(T, R, MULTI);
}

template <int T, int R, bool MULTI, int DENSE = 0, bool CKL = false>
__global__ __launch_bounds__(DENSE ? RES_DENSE_BLOCK : Team<T>::BLOCK)
    __attribute__((amdgpu_waves_per_eu(res_min_waves_dense(T, R, MULTI, DENSE)))) void k_nuts_resident(EngineArgs a, NutsSampleArgs m) {
  using TM = Team<T>;
  static_assert(DENSE == 0 || (T == 64 && R == 1), "dense products: one wavefront per chain, one element per lane");
  static_assert(!CKL || (DENSE == 0 && T == 64 && R == 1), "LDS checkpoints: one wavefront per chain, one element per lane");
  constexpr bool MD = (DENSE & RES_DENSE_METRIC) != 0, TD = (DENSE & RES_DENSE_TARGET) != 0;
  constexpr bool PC = (DENSE & RES_DENSE_PER_CHAIN) != 0, MLDS = MD && !PC;
  constexpr int BLOCK = DENSE ? RES_DENSE_BLOCK : TM::BLOCK;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c = TM::SUB ? ((long long)blockIdx.x * 256 + threadIdx.x) / T : (long long)blockIdx.x * (BLOCK / 64) + wave;
  const int t = TM::SUB ? (int)(threadIdx.x % T) : lane;
  // DENSE: the shared matrices, transposed, in LDS (immT: the inverse mass matrix, smT: L^-T, PT: the precision)
  extern __shared__ __attribute__((aligned(16))) double res_lds[];
  const int Dd = (int)a.D, DD = Dd * Dd;
  const double *immW = res_lds, *const smT = res_lds + (MLDS ? DD : 0), *const PT = res_lds + (MLDS ? 2 * DD : 0);
  if (DENSE) {
    for (int e = threadIdx.x; e < DD; e += BLOCK) {
      const int i = e / Dd, k = e % Dd;
      if (MLDS) {
        res_lds[k * Dd + i] = a.imm[e];
        res_lds[DD + k * Dd + i] = a.sqrt_mass[e];
      }
      if (TD) res_lds[(MLDS ? 2 * DD : 0) + k * Dd + i] = m.prec[e];
    }
    __syncthreads();
  }
  if (c >= a.C) return;  // a whole team leaves together
  if (PC) {  // this chain's transposed inverse mass matrix (workspace; rewritten when the metric may have changed)
    wave_transpose_to(a.imm + (size_t)c * DD, m.imm_ws + (size_t)c * DD, Dd, lane);
    immW = m.imm_ws + (size_t)c * DD;
  }
  double v0 = 0.0;  // MD: the moving end's velocity imm p (element `lane`)
  const size_t row = (size_t)c * a.D;
  const bool lead = t == 0;
  const bool single = T == 64 && a.D == 1;  // (wave-uniform: a kernel argument)

  double q[R], g[R], p[R], pb[R];  // moving end + sub-trajectory momentum sum
  bool ok[R];
// large R: keep the scheduler from interleaving all R unrolled iterations (their
// temporaries would not fit the register file and spill)
// element r of this thread in a [C,D] array: uniform row base (SGPRs) + 32-bit lane offset,
// so one VGPR offset per r serves every array (64-bit per-array addresses would not fit)
#define EI(r) ((unsigned)(t + T * (r)))
#define AT(ptr, r) ((ptr) + row)[EI(r)]
#define R_FENCE() do { if (R > 4) __builtin_amdgcn_sched_barrier(0); } while (0)
#define QGET(r) (q[r])
#define GGET(r) (g[r])
#define QSET(r, v) do { q[r] = (v); } while (0)
#define GSET(r, v) do { g[r] = (v); } while (0)
  // imm is re-read from L1/L2 when many elements per thread would cost registers
  constexpr bool IM_REG = R <= 4;
  constexpr int BR = R > 4 ? 4 : R;  // elements whose global operands are in flight together
  double imr[IM_REG ? R : 1];
  const size_t imo = (size_t)c * a.imm_cs;
#define IMM(r) (IM_REG ? imr[IM_REG ? (r) : 0] : a.imm[imo + (a.met_ndim == 0 ? 0 : (long long)t + (long long)T * (r))])
  ChainRng rng = {};
  ChainCtl ct = {};
  double kd = 0.0, zero = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const long long i = (long long)t + (long long)T * r;
    ok[r] = i < a.D;
    if (IM_REG) imr[r] = (ok[r] && !MD) ? a.imm[imo + (a.met_ndim == 0 ? 0 : i)] : 1.0;
  }
  rng = rng_load(a, c);
  double U_state = MULTI ? a.U[c] : 0.0;  // the chain's potential energy between transitions
  long long nleap_sum = 0;
  // window adaptation inside the launch (m.adapt; window_adaptation.py:17-116, diagonal mass matrix, one
  // row of the state arrays per chain): the team's scalars in registers, Welford sums / metric in memory
  DualAvg da = {1, 0.0, 0.0, 0.0, 0.0};
  long long wc_n = 0;
  double eps_adapt = 0.0;
  if (MULTI && m.adapt) {
    da.step = m.ad.da_step[c];
    da.x = m.ad.da_x[c];
    da.x_avg = m.ad.da_x_avg[c];
    da.g_avg = m.ad.da_g_avg[c];
    da.mu = m.ad.da_mu[c];
    wc_n = m.ad.wc_n[c];
    eps_adapt = m.ad.step_size[c];
  }
  // m.T consecutive transitions in this launch (the user-level scan of tests/test_hmc.py:296-324): the
  // chains of a wavefront start each transition together, wavefronts run independently of each other
  for (long long t_idx = 0; t_idx < (MULTI ? m.T : 1); t_idx++) {
  {
#pragma unroll
  for (int r = 0; r < R; r++) {
    QSET(r, ok[r] ? AT(a.q, r) : 0.0);  // (after the first transition: what this thread wrote below)
    GSET(r, ok[r] ? AT(a.g, r) : 0.0);
  }

  // ---- momentum, site #1 (nuts.py:113 -> metrics.py:65-68) ------------------------
  if (TM::SUB) {
    // every lane of the team walks the chain's stream itself and keeps its own elements
    const double *sm = a.sqrt_mass + imo;
#pragma unroll
    for (int r = 0; r < R; r++) {
      p[r] = 0.0;
      for (int tt = 0; tt < T; tt++) {
        const long long i = (long long)tt + (long long)T * r;
        if (i < a.D) {
          const double z = rng_standard_normal(rng.g[0]);
          if (tt == t) p[r] = (a.met_ndim == 0 ? sm[0] : sm[i]) * z;
        }
      }
    }
    if (!MULTI && lead) pcg_store(a.rng + (size_t)c * a.nsites * 4, rng.g[0]);
  } else {
    {
      const double *sm = a.sqrt_mass + imo;
      const bool scalar = a.met_ndim == 0;
      double *dst = a.zbuf;
      if (MD) wave_normals(rng.g[0], a.D, [=](long long i, double z) { dst[row + i] = z; });  // p = L^-T z below
      else wave_normals(rng.g[0], a.D, [=](long long i, double z) { dst[row + i] = (scalar ? sm[0] : sm[i]) * z; });
      if (!MULTI && lead) pcg_store(a.rng + (size_t)c * a.nsites * 4, rng.g[0]);
    }
    __threadfence_block();
  }

  // ---- nuts.py:113-125 ----------------------------------------------------------------
  kd = 0.0;
  if (MD) {  // metrics.py:66-67 and :71
    const double z = ok[0] ? AT(a.zbuf, 0) : 0.0;
    p[0] = PC ? wave_matvec_rows_reg(a.sqrt_mass + (size_t)c * DD, z, Dd, lane) : wave_matvec_reg(smT, z, Dd, lane);
    if (!ok[0]) p[0] = 0.0;
    v0 = wave_matvec_reg(immW, p[0], Dd, lane);
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (!TM::SUB && !MD) p[r] = ok[r] ? AT(a.zbuf, r) : 0.0;
    pb[r] = 0.0;
    if (ok[r]) {
      kd += MD ? v0 * p[r] : (IMM(r) * p[r]) * p[r];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        AT(a.end_q[e], r) = QGET(r);
        AT(a.end_p[e], r) = p[r];
        AT(a.end_g[e], r) = GGET(r);
        if (MD) AT(a.end_v[e], r) = v0;
      }
      AT(a.slot_q[0], r) = QGET(r);
      AT(a.slot_p[0], r) = p[r];
      AT(a.slot_g[0], r) = GGET(r);
      AT(a.psum, r) = p[r];
    }
  }
  team_sum2<T>(kd, zero, single);
  {
    const double U = MULTI ? U_state : a.U[c];
    ct.H0 = U + 0.5 * kd;
    ct.prop_E = ct.H0;
    ct.prop_w = 0.0;
    ct.prop_slpa = -INFINITY;
    ct.sub_E = ct.sub_w = ct.sub_slpa = 0.0;
    ct.U_cur = ct.U_end[0] = ct.U_end[1] = ct.U_slot[0] = ct.U_slot[1] = U;
    ct.acc_prob = 0.0;
    ct.nleap = 0;
    ct.j = 0;
    ct.length = 0;
    ct.tmin = ct.tmax = 0;
    ct.done = ct.phantom = 0;
    ct.prop_slot = 0;
    ct.ndoubl = ct.out_div = ct.out_turn = 0;
    ct.dir = rng_bernoulli(rng.g[1], 0.5);  // trajectory.py:516
    ct.step = 0;
  }
  }
  const double eps = (MULTI && m.adapt) ? eps_adapt : (a.eps_c ? a.eps_c[c] : a.eps);

  while (!ct.done) {
    // ---- one leapfrog of the moving end, in registers (integrators.py:54-73) ---------
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size, aa = 1 * step_size;
    double usum = 0.0;
    kd = 0.0;
    if (DENSE) {  // products between the stages: engine.cuh's leap_small_dense
      double qq = QGET(0), gg = GGET(0);
      ct.U_cur = leap_small_dense<MD, TD>(a, c, lane, step_size, immW, PT, Dd, qq, p[0], gg);
      QSET(0, qq);
      GSET(0, gg);
      if (MD) v0 = wave_matvec_reg(immW, p[0], Dd, lane);  // imm p'
      kd = ok[0] ? (MD ? v0 * p[0] : (IMM(0) * p[0]) * p[0]) : 0.0;
      kd = wave_sum(kd);
    }
#ifdef AEHMC_JOINT_GRAD
    else if (T == 64 && a.tkind == AEHMC_T_JOINT) {
      // a joint density with its reverse-mode program (round 6), 64 < D <= 512: the chain in registers as for the
      // coordinate-wise targets, the position handed to the program through the wavefront's LDS rows -- first stages |
      // q' -> LDS | logp and its gradient in one sweep | dU/dq' <- LDS, last stage.  The leapfrog's arithmetic is the
      // loop's below (= engine.cuh leap_stages): the same bits as k_nuts_joint_rows, without its passes over L2 rows.
      double *const qrow = res_lds + (size_t)wave * 2 * a.D, *const grow = qrow + a.D;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double pp = p[r] - b * GGET(r);
          const double qq = QGET(r) + aa * (IMM(r) * pp);
          QSET(r, qq);
          p[r] = pp;
          qrow[EI(r)] = qq;
          grow[EI(r)] = 0.0;
        }
      }
      __threadfence_block();  // (the rows are read and updated through other lanes' addresses)
      const double lp = aehmc_logp_grad(qrow, grow, lane, a.cparams);
      __threadfence_block();
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double gg = -grow[EI(r)];
          const double pp = p[r] - b * gg;
          GSET(r, gg);
          p[r] = pp;
          kd += (IMM(r) * pp) * pp;
        }
      }
      team_sum2<T>(usum, kd, single);
      ct.U_cur = -lp;
    }
#endif
    else {
      // (global operands -- imm when it is not in registers -- are fetched BR elements at a
      // time: one round trip per batch instead of one per element)
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += BR) {
        double imv[BR];
#pragma unroll
        for (int u = 0; u < BR; u++) imv[u] = (r0 + u < R && ok[r0 + u < R ? r0 + u : 0]) ? IMM(r0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < BR; u++) {
          const int r = r0 + u;
          if (r < R && ok[r < R ? r : 0]) {
            const long long i = (long long)t + (long long)T * r;
            double pp = p[r] - b * GGET(r);
            double qq = QGET(r) + aa * (imv[u] * pp);
            double uu, gg;
            target_elem(a, i, qq, uu, gg);
            usum += uu;
            pp = pp - b * gg;
            QSET(r, qq);
            GSET(r, gg);
            p[r] = pp;
            kd += (imv[u] * pp) * pp;
          }
        }
        R_FENCE();
      }
      team_sum2<T>(usum, kd, single);
      ct.U_cur = target_finish(a, usum);
    }

    // ---- dynamic_integration body (trajectory.py:195-305) ------------------------------
    const int step = ct.step;
    if (!ct.phantom) ct.nleap += 1;
    const TreeIdx ti = tree_step_indices(step, ct.tmin, ct.tmax);  // termination.py:109-113 (stale at step 0), 192-235
    const int tmin = ti.tmin, tmax = ti.tmax;
    const bool even = (step & 1) == 0;
    {
      double *ckp = CKL ? res_lds + ((size_t)(wave * a.max_exp + tmax) * 2) * 64 : a.ckp + ((size_t)tmax * a.C + c) * a.D;
      double *cks = CKL ? res_lds + ((size_t)(wave * a.max_exp + tmax) * 2 + 1) * 64 : a.cks + ((size_t)tmax * a.C + c) * a.D;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          pb[r] = (step == 0) ? p[r] : pb[r] + p[r];
          if (even) {
            ckp[EI(r)] = p[r];
            cks[EI(r)] = pb[r];
            if (MD) (a.ckv + ((size_t)tmax * a.C + c) * a.D)[EI(r)] = v0;
          }
        }
        R_FENCE();
      }
    }
    ct.tmin = tmin;
    ct.tmax = tmax;
    // proposals.py:19-62, 72-102, 141-144 (nuts_tree.cuh; a wavefront that owns one chain evaluates the three
    // transcendental chains of a step in three lanes at once, sub-wavefront teams in scalar form)
    const TreePoint np = tree_new_point(ct.H0, ct.U_cur, kd, a.thr);
    const bool div = np.div;
    bool term = false;
    const bool take = tree_sample_step<TM::WAVE>(ct, step, np, lane, [&](double pr) { return rng_bernoulli(rng.g[2], pr); });
    if (step >= 1) {
      if (tmax >= tmin) {  // termination.py:133-187
        int idx = tmax;
        bool crit = false;
        for (;;) {
          const double *kp = CKL ? res_lds + ((size_t)(wave * a.max_exp + idx) * 2) * 64 : a.ckp + ((size_t)idx * a.C + c) * a.D;
          const double *ks = CKL ? res_lds + ((size_t)(wave * a.max_exp + idx) * 2 + 1) * 64 : a.cks + ((size_t)idx * a.C + c) * a.D;
          const double *kv = a.ckv + ((size_t)idx * a.C + c) * a.D;  // (MD)
          double d_l = 0.0, d_r = 0.0;
#pragma unroll
          for (int r0 = 0; r0 < R; r0 += BR) {
            double kpv[BR], ksv[BR], imv[BR];
#pragma unroll
            for (int u = 0; u < BR; u++) {
              const int r = r0 + u < R ? r0 + u : 0;
              const bool on = r0 + u < R && ok[r];
              kpv[u] = on ? kp[EI(r)] : 0.0;
              ksv[u] = on ? ks[EI(r)] : 0.0;
              imv[u] = on ? (MD ? kv[EI(r)] : IMM(r)) : 0.0;  // MD: the checkpoint's velocity itself
            }
#pragma unroll
            for (int u = 0; u < BR; u++) {
              const int r = r0 + u < R ? r0 + u : 0;
              if (r0 + u < R && ok[r]) {
                double pl = kpv[u], pr = p[r];
                double vl = MD ? imv[u] : imv[u] * pl, vr = MD ? v0 : imv[u] * pr;
                double sub = pb[r] - ksv[u] + pl;
                double rho = sub - (pr + pl) / 2;
                d_l += vl * rho;
                d_r += vr * rho;
              }
            }
            R_FENCE();
          }
          team_sum2<T>(d_l, d_r, single);
          crit = (d_l <= 0) | (d_r <= 0);
          bool reached = (idx - 1) < tmin;
          idx -= 1;
          if (crit || reached) break;
        }
        term = crit;
      }
    }
    if (take) {  // sub-trajectory proposal <- moving end (copy on accept)
      const int s = ct.prop_slot ^ 1;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          AT(pick2(a.slot_q, s), r) = QGET(r);
          AT(pick2(a.slot_p, s), r) = p[r];
          AT(pick2(a.slot_g, s), r) = GGET(r);
        }
        R_FENCE();
      }
      put2(ct.U_slot, s, ct.U_cur);
    }

    // ---- sub-trajectory / expansion control (trajectory.py:336, 537-608) ----------------
    const TreeControl tc = tree_step_control(ct, step, div, term);
    const bool fin_div = tc.fin_div, fin_term = tc.fin_term;
    if (tc.finalize) {
      const int dir = ct.dir, oth = 1 - dir;
      double d_l = 0.0, d_r = 0.0;
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += BR) {
        double pov[BR], psv[BR], imv[BR];
#pragma unroll
        for (int u = 0; u < BR; u++) {
          const int r = r0 + u < R ? r0 + u : 0;
          const bool on = r0 + u < R && ok[r];
          pov[u] = on ? AT(pick2(a.end_p, oth), r) : 0.0;
          psv[u] = on ? AT(a.psum, r) : 0.0;
          imv[u] = on ? (MD ? AT(pick2(a.end_v, oth), r) : IMM(r)) : 0.0;  // MD: the other end's velocity
        }
#pragma unroll
        for (int u = 0; u < BR; u++) {
        const int r = r0 + u < R ? r0 + u : 0;
        if (r0 + u < R && ok[r]) {
          double pc = p[r], po = pov[u];
          double vc = MD ? v0 : imv[u] * pc, vo = MD ? imv[u] : imv[u] * po;
          double s = psv[u] + pb[r];
          AT(a.psum, r) = s;
          double pl = dir ? po : pc, pr = dir ? pc : po;
          double vl = dir ? vo : vc, vr = dir ? vc : vo;
          double rho = s - (pr + pl) / 2;
          d_l += vl * rho;
          d_r += vr * rho;
          AT(pick2(a.end_q, dir), r) = QGET(r);
          AT(pick2(a.end_p, dir), r) = pc;
          AT(pick2(a.end_g, dir), r) = GGET(r);
          if (MD) AT(pick2(a.end_v, dir), r) = vc;
        }
        }
        R_FENCE();
      }
      team_sum2<T>(d_l, d_r, single);
      const bool turning = (d_l <= 0) | (d_r <= 0);
      put2(ct.U_end, dir, ct.U_cur);
      if (tree_merge_expansion<TM::WAVE>(ct, fin_div, fin_term, lane, [&](double pr) { return rng_bernoulli(rng.g[3], pr); }))
        ct.prop_slot ^= 1;
      const bool end_transition = tree_expansion_outcome(ct, fin_div, fin_term, turning, a.max_exp);
      if (end_transition) {
        const int s = ct.prop_slot;  // outputs (the phantom scan below cannot change them)
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (ok[r]) {
            AT(a.q, r) = AT(pick2(a.slot_q, s), r);
            AT(a.g, r) = AT(pick2(a.slot_g, s), r);
            if (a.out.momentum) AT(a.out.momentum, r) = AT(pick2(a.slot_p, s), r);
          }
        }
        if (MULTI) U_state = pick2(ct.U_slot, s);
        if (lead) {
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
            if (ok[r]) {
              QSET(r, AT(pick2(a.end_q, go_right), r));
              p[r] = AT(pick2(a.end_p, go_right), r);
              GSET(r, AT(pick2(a.end_g, go_right), r));
              if (MD) v0 = AT(pick2(a.end_v, go_right), r);
            }
          }
          ct.U_cur = pick2(ct.U_end, go_right);
        }
      }
    }
  }
  // per-transition records of a multi-transition launch
  if (MULTI) nleap_sum += ct.nleap;
  if (MULTI && m.samples) {
#pragma unroll
    for (int r = 0; r < R; r++)
      if (ok[r]) (m.samples + ((size_t)t_idx * a.C + c) * a.D)[EI(r)] = AT(a.q, r);
  }
  if (MULTI && lead) {
    if (m.acc_hist) m.acc_hist[(size_t)t_idx * a.C + c] = ct.acc_prob;
    if (m.div_hist) m.div_hist[(size_t)t_idx * a.C + c] = ct.out_div;
  }
  if (MULTI && m.adapt) {  // the update of k_adapt_update, same order: step size, Welford, window end, last
    const int stage = m.stage[t_idx], wend = m.window_end[t_idx];
    double step_size = adapt_da_update(da, m.target, ct.acc_prob, m.gamma, m.t0, m.kappa);
    if (MD && PC) {
      // is_mass_matrix_full: the arithmetic of k_adapt_update's full branch (algorithms.py:187-197 with np.outer,
      // mass_matrix.py:83-118), element i of the position in lane i, the D x D Welford sum row by row in memory
      double *const m2 = m.ad.wc_m2 + (size_t)c * DD;
      if (stage != 0) {
        wc_n += 1;
        double delta = 0.0, ud = 0.0;
        if (ok[0]) {
          const double v = AT(a.q, 0);
          double mean = AT(m.ad.wc_mean, 0);
          delta = v - mean;
          mean = mean + delta / (double)wc_n;
          AT(m.ad.wc_mean, 0) = mean;
          ud = v - mean;
        }
        for (int i = 0; i < Dd; i++) {
          const double ud_i = read_lane_f64(ud, i);
          if (ok[0]) m2[i * Dd + lane] = m2[i * Dd + lane] + ud_i * delta;
        }
      }
      if (wend) {
        const double nn = (double)wc_n;
        double *const Aw = m.imm_ws + (size_t)c * DD;  // scratch of the factorisation (then the transposed copy again)
        __threadfence_block();
        for (int idx = lane; idx < DD; idx += 64) {
          const double cov = m2[idx] / (double)(wc_n - 1);
          double im = (nn / (nn + 5)) * cov;
          if (idx / Dd == idx % Dd) im = im + 1e-3 * (5 / (nn + 5));  // shrinkage * eye
          m.ad.imm[(size_t)c * DD + idx] = im;  // (== a.imm: the metric bound to this call)
          Aw[idx] = im;
          m2[idx] = 0.0;
        }
        if (ok[0]) AT(m.ad.wc_mean, 0) = 0.0;
        __threadfence_block();
        wave_chol_inv_t(Aw, m.ad.sqrt_mass + (size_t)c * DD, Dd, lane);  // a non-PD estimate leaves NaNs (as the reference)
        wave_transpose_to(a.imm + (size_t)c * DD, Aw, Dd, lane);
        wc_n = 0;
        adapt_da_restart(da, step_size);
      }
    } else {
    if (stage != 0) {
      wc_n += 1;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (ok[r]) {
          double mean = AT(m.ad.wc_mean, r), m2 = AT(m.ad.wc_m2, r);
          adapt_welford_elem(AT(a.q, r), wc_n, mean, m2);
          AT(m.ad.wc_mean, r) = mean;
          AT(m.ad.wc_m2, r) = m2;
        }
    }
    if (wend) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if (ok[r]) {
          double mean = AT(m.ad.wc_mean, r), m2 = AT(m.ad.wc_m2, r), imm, sqrt_mass;
          adapt_window_end_elem(wc_n, mean, m2, imm, sqrt_mass);
          AT(m.ad.imm, r) = imm;  // (== a.imm / a.sqrt_mass: the metric bound to this call)
          AT(m.ad.sqrt_mass, r) = sqrt_mass;
          AT(m.ad.wc_mean, r) = mean;
          AT(m.ad.wc_m2, r) = m2;
          if (IM_REG) imr[IM_REG ? r : 0] = imm;
        }
      wc_n = 0;
      adapt_da_restart(da, step_size);
      __threadfence_block();  // the next momentum draw reads other lanes' sqrt_mass elements
    }
    }
    if (t_idx == m.T - 1) step_size = exp(da.x_avg);  // window_adaptation.py:184-190
    eps_adapt = step_size;
  }
  }  // transitions
  if (lead) {
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = ct.nleap;
    if (MULTI && m.nleap_total) m.nleap_total[c] = nleap_sum;
    if (MULTI && m.adapt) {
      m.ad.da_step[c] = da.step;
      m.ad.da_x[c] = da.x;
      m.ad.da_x_avg[c] = da.x_avg;
      m.ad.da_g_avg[c] = da.g_avg;
      m.ad.da_mu[c] = da.mu;
      m.ad.wc_n[c] = wc_n;
      m.ad.step_size[c] = eps_adapt;
    }
    if (MULTI) pcg_store(a.rng + ((size_t)c * a.nsites + 0) * 4, rng.g[0]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, rng.g[1]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 2) * 4, rng.g[2]);
    pcg_store(a.rng + ((size_t)c * a.nsites + 3) * 4, rng.g[3]);
  }
#undef IMM
#undef EI
#undef AT
#undef R_FENCE
#undef QGET
#undef GGET
#undef QSET
#undef GSET
}

inline bool nuts_resident_supported(int tkind, int met_ndim, long long D) {
  // teams of 1 .. 64 lanes up to D = 512; larger chains take one workgroup each (nuts_wide.cuh)
  return (tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN) &&
         met_ndim < 2 && D <= 512;
}

inline bool nuts_resident_dense_supported(int tkind, int met_ndim, long long D) {
  const bool elem = tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN;
  return D <= FUSED_DENSE_MAX_D && (met_ndim == 2 || tkind == AEHMC_T_DENSE_MVN) && (elem || tkind == AEHMC_T_DENSE_MVN);
}
#ifndef __HIPCC_RTC__
template <int DENSE>
inline hipError_t launch_nuts_resident_dense_v(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) {
  constexpr int nmat = ((DENSE & RES_DENSE_METRIC) && !(DENSE & RES_DENSE_PER_CHAIN) ? 2 : 0) +
                       ((DENSE & RES_DENSE_TARGET) ? 1 : 0);
  const size_t dyn = (size_t)nmat * a.D * a.D * sizeof(double);
  const unsigned grid = (unsigned)((a.C + RES_DENSE_BLOCK / 64 - 1) / (RES_DENSE_BLOCK / 64));
  const bool multi = m.T > 1 || m.samples || m.acc_hist || m.div_hist || m.nleap_total || m.adapt;
  // in-launch adaptation of dense matrices: one per chain, with the transposed-copy workspace as factorisation scratch
  if (m.adapt && !((DENSE & RES_DENSE_METRIC) && (DENSE & RES_DENSE_PER_CHAIN) && m.ad.full && m.imm_ws))
    return hipErrorInvalidValue;
#define AEHMC_RD(MULTI)                                                                                      \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_resident<64, 1, MULTI, DENSE>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_nuts_resident<64, 1, MULTI, DENSE>), dim3(grid), dim3(RES_DENSE_BLOCK), dyn, st, a, m); \
  } while (0)
  if (multi) AEHMC_RD(true);
  else AEHMC_RD(false);
#undef AEHMC_RD
  return hipGetLastError();
}
// md / td: dense metric / dense-precision target; pc: one dense metric per chain (m.imm_ws: [C, D, D] workspace)
inline hipError_t launch_nuts_resident_dense(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st, bool md,
                                             bool td, bool pc) {
  if (md && td && pc) return launch_nuts_resident_dense_v<7>(a, m, st);
  if (md && td) return launch_nuts_resident_dense_v<3>(a, m, st);
  if (md && pc) return launch_nuts_resident_dense_v<5>(a, m, st);
  if (md) return launch_nuts_resident_dense_v<1>(a, m, st);
  if (td) return launch_nuts_resident_dense_v<2>(a, m, st);
  return hipErrorInvalidValue;
}
#endif  // __HIPCC_RTC__

// Which instantiation a call takes.  Team size: the smallest team that holds the chain (<= 4 elements per lane below a
// wave, <= 8-10 above), widened -- fewer chains per wavefront -- while that still leaves about 4096 wavefronts in
// flight, because a wider team wastes lanes but keeps the per-chain RNG / control state wave-uniform (SGPRs, scalar
// branches).  CKL: a few chains (at most two wavefronts per SIMD) keep their checkpoints in LDS, 4 waves x max_exp
// levels x 1 KB <= 64 KB.
struct ResidentPlan {
  int T, R;
  bool multi, ckl;
  unsigned grid;
  size_t dyn;
};
inline ResidentPlan plan_nuts_resident(const EngineArgs &a, const NutsSampleArgs &m, int force_min_team = 0) {
  const long long D = a.D, C = a.C;
  const int tmin = D <= 4 ? 1 : D <= 8 ? 2 : D <= 16 ? 4 : D <= 32 ? 8 : D <= 64 ? 16 : D <= 128 ? 32 : 64;
  int twant = 64;
  while (twant > 1 && C * (twant / 2) >= 64LL * 4096) twant /= 2;
  ResidentPlan p{};
  p.T = force_min_team ? tmin : (tmin > twant ? tmin : twant);
  if (p.T == 1) p.R = D <= 1 ? 1 : D <= 2 ? 2 : 4;
  else if (p.T < 64) p.R = 4;
  else p.R = D <= 64 ? 1 : D <= 128 ? 2 : D <= 256 ? 4 : 8;
  p.multi = m.T > 1 || m.samples || m.acc_hist || m.div_hist || m.nleap_total || m.adapt;
  p.ckl = p.T == 64 && p.R == 1 && a.C <= 2048 && a.max_exp <= 16;
  p.grid = p.T < 64 ? (unsigned)((a.C * p.T + 255) / 256) : (unsigned)((a.C + 3) / 4);
  p.dyn = p.ckl ? (size_t)4 * a.max_exp * 2 * 64 * sizeof(double) : 0;
  return p;
}

#ifndef __HIPCC_RTC__
template <int T, int R>
inline hipError_t launch_nuts_resident_tr(const EngineArgs &a, const NutsSampleArgs &m, const ResidentPlan &p,
                                          hipStream_t st) {
  if constexpr (T == 64 && R == 1) {
    if (p.ckl) {
      if (p.multi) hipLaunchKernelGGL((k_nuts_resident<T, R, true, 0, true>), dim3(p.grid), dim3(Team<T>::BLOCK), p.dyn, st, a, m);
      else hipLaunchKernelGGL((k_nuts_resident<T, R, false, 0, true>), dim3(p.grid), dim3(Team<T>::BLOCK), p.dyn, st, a, m);
      return hipGetLastError();
    }
  }
  if (p.multi) hipLaunchKernelGGL((k_nuts_resident<T, R, true>), dim3(p.grid), dim3(Team<T>::BLOCK), 0, st, a, m);
  else hipLaunchKernelGGL((k_nuts_resident<T, R, false>), dim3(p.grid), dim3(Team<T>::BLOCK), 0, st, a, m);
  return hipGetLastError();
}
inline hipError_t launch_nuts_resident(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st,
                                       int force_min_team = 0) {
  if (a.D > 512) return hipErrorInvalidValue;  // one workgroup per chain: nuts_wide.cuh
  const ResidentPlan p = plan_nuts_resident(a, m, force_min_team);
  switch (p.T) {
    case 1:
      if (p.R == 1) return launch_nuts_resident_tr<1, 1>(a, m, p, st);
      if (p.R == 2) return launch_nuts_resident_tr<1, 2>(a, m, p, st);
      return launch_nuts_resident_tr<1, 4>(a, m, p, st);
    case 2: return launch_nuts_resident_tr<2, 4>(a, m, p, st);
    case 4: return launch_nuts_resident_tr<4, 4>(a, m, p, st);
    case 8: return launch_nuts_resident_tr<8, 4>(a, m, p, st);
    case 16: return launch_nuts_resident_tr<16, 4>(a, m, p, st);
    case 32: return launch_nuts_resident_tr<32, 4>(a, m, p, st);
    default:
      if (p.R == 1) return launch_nuts_resident_tr<64, 1>(a, m, p, st);
      if (p.R == 2) return launch_nuts_resident_tr<64, 2>(a, m, p, st);
      if (p.R == 4) return launch_nuts_resident_tr<64, 4>(a, m, p, st);
      return launch_nuts_resident_tr<64, 8>(a, m, p, st);
  }
}
#endif  // __HIPCC_RTC__

}  // namespace aehmc
