// Device side of the lock-step many-chain HMC/NUTS engine (gfx950, wave64).
//
// One wavefront owns one chain.  Vectors are rows of chain-major [C,D] fp64 arrays in
// HBM (coalesced 512 B per wave access); per-chain scalars live in a ChainCtl record.
// A NUTS transition is a per-chain state machine advanced by one leapfrog per launch
// ("lock-step"): chains at different tree depths / directions share each launch, and
// with a dense metric or dense target the mat-vecs of all chains become one fp64 MFMA
// GEMM between the stages (gemm_f64.cuh).
//
// Reference semantics restated here (file:line under /root/reference/aehmc):
//   leapfrog stages S1..S3   integrators.py:54-73
//   kinetic energy / U-turn   metrics.py:70-104
//   proposal scalars          proposals.py:19-62, 72-174
//   checkpoint bookkeeping    termination.py:85-235
//   sub-trajectory loop       trajectory.py:154-374
//   expansion loop            trajectory.py:428-714, nuts.py:56-153
//   HMC accept/reject         hmc.py:157-204
#pragma once
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime, the math functions and the fixed-width integers itself) */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#endif

#include "../../include/aehmc_hip.h"
#include "linreg_rows.cuh"
#include "rng.cuh"

namespace aehmc {

#define AEHMC_LOG_SQRT_2PI 0.91893853320467267

struct ChainCtl {
  double H0;                        // initial energy of the transition
  double prop_E, prop_w, prop_slpa; // main proposal scalars (proposals.py:11-15)
  double sub_E, sub_w, sub_slpa;    // sub-trajectory proposal scalars
  double U_cur, U_end[2], U_slot[2];
  double acc_prob;
  long long nleap;
  int j, step, dir, length;         // expansion, step in sub-trajectory, 1 = right, length
  int tmin, tmax;                   // termination.py:12-16 indices (carried, never reset)
  int done, phantom;                // phantom: first step diverged, scan still runs (trajectory.py:336)
  int prop_slot;                    // which of the two proposal buffers is the main proposal
  int ndoubl, out_div, out_turn;
};

// Occupancy bound of the kernels that run a USER's density in workgroups of more than 256 threads.  A 64 W-thread workgroup
// needs W / 4 wavefronts per SIMD resident together, i.e. at most 512 / (W / 4) registers per lane, accumulation registers
// included.  __launch_bounds__ alone, and amdgpu_waves_per_eu(W / 4) too, let the compiler of hipRTC 7.2 go past that for
// densities with long unrolled bodies (236 + 32 accumulation registers = 268 for a 512-thread workgroup, 328 with
// amdgpu_waves_per_eu(1)): a code object that cannot be launched -- HSA_STATUS_ERROR_INVALID_ISA, the process aborts.
// One wavefront more is asked for (W = 8: three per SIMD, <= 168 + accumulation registers; measured 228).
// (AEHMC_WG_MIN_WAVES: the engine compiles the eight-wavefront kernels for FOUR per SIMD first -- two workgroups per CU -- and
//  falls back to three when that program spills: engine.hip wg_program.)
#ifdef AEHMC_WG_MIN_WAVES
constexpr int wg_min_waves(int W) { return W == 8 ? AEHMC_WG_MIN_WAVES : (W >= 4 ? W / 4 : 1) + 1; }
#else
constexpr int wg_min_waves(int W) { return (W >= 4 ? W / 4 : 1) + 1; }
#endif
#ifdef AEHMC_JOINT_TARGET
#define AEHMC_USER_DENSITY_512 __attribute__((amdgpu_waves_per_eu(3)))  /* (512-thread kernels compiled against a user's joint density) */
#else
#define AEHMC_USER_DENSITY_512
#endif

struct EngineArgs {
  long long C, D;
  long long ldw;        // row stride of the work arrays: D, or D rounded up for the workgroup-per-chain NUTS kernel
  double eps, thr;
  const double *eps_c;  // optional per-chain step sizes [C] (window adaptation is per chain)
  int max_exp;
  // metric (metrics.py:44-63); imm_cs: chain stride of imm / sqrt_mass (0 = shared)
  int met_ndim;
  long long imm_cs;
  const double *imm, *sqrt_mass;
  // target
  int tkind;
  const double *mu, *sigma, *log_sigma;
  const double *X, *y;  // linreg data [N]
  long long N;
  const double *const *cparams;  // user-defined target (AEHMC_T_CUSTOM): device array of its parameter arrays
  // per-chain RNG [C, nsites, 4]
  uint64_t *rng;
  int nsites;
  // work vectors [C,D]
  double *cur_q, *cur_p, *cur_g, *cur_v, *cur_w;
  double *end_q[2], *end_p[2], *end_g[2], *end_v[2], *end_w[2];
  double *slot_q[2], *slot_p[2], *slot_g[2];
  double *psum, *psub;
  double *ckp, *cks, *ckv;          // [max_exp][C][D]
  double *vhalf, *rbuf, *zbuf;
  double *linreg_part;  // [ceil(C/8)][S][16] slice sums of the regression target
  ChainCtl *ctl;
  // dense metric, "linear" mode: w = imm g is carried with the state so that
  // v_half = v - (eps/2) w and v' = v_half - (eps/2) w' need one metric GEMM per leapfrog
  int linear;
  // compacted list of live chains for the GEMMs (built by k_compact)
  int *row_idx, *n_rows;
  // caller state / outputs
  double *q, *U, *g;
  aehmc_diagnostics out;
};

// optional per-transition outputs of a launch that runs T transitions
struct NutsSampleArgs {
  long long T;
  double *samples;         // [T][C][D] positions after each transition
  double *acc_hist;        // [T][C]
  int *div_hist;           // [T][C]
  long long *nleap_total;  // [C] leapfrogs of all T transitions
  // window adaptation inside the launch (window_adaptation.py:17-116, diagonal mass matrix): after
  // its transition t a chain updates its own dual-averaging / Welford state with schedule entry t
  // and goes on with the new step size (and, after a window end, the new metric)
  int adapt;
  const int *stage, *window_end;  // device arrays [T] (window_adaptation.py:230-327)
  double target, gamma, t0, kappa;
  aehmc_adapt_state ad;
  // small dense problems (k_nuts_resident's DENSE instantiations)
  const double *prec;  // the dense target's precision [D, D]
  double *imm_ws;      // per-chain dense metrics: [C, D, D] workspace for the transposed matrices
  // block-resident kernels (nuts_block_reg.cuh): waiting chains of a workgroup begin their next transition once this
  // many wait (0: the kernel's default; 16: all together, transition by transition)
  int roll;
};

// two-entry arrays are picked with a select, never indexed dynamically (a dynamic index
// into the kernel-argument struct or ChainCtl sends them to scratch memory)
template <class T>
__device__ __forceinline__ T pick2(T const (&arr)[2], int i) {
  return i ? arr[1] : arr[0];
}
__device__ __forceinline__ void put2(double (&arr)[2], int i, double v) {
  if (i) arr[1] = v;
  else arr[0] = v;
}

// Wavefront all-reduce of a double.  Four DPP butterfly stages (xor 1, xor 2, mirror within
// 8, mirror within 16: every stage adds a value to its mirror image, so all lanes of a
// 16-lane row end with the same bits), then the four row sums are read with v_readlane and
// added in a fixed order.  The result is wave-uniform by construction (and known to be so
// by the compiler, which keeps it in SGPRs) -- no LDS-crossbar ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return x + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double read_lane_f64(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                          __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double wave_sum(double x) {
  x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);  // row_half_mirror
  x = dpp_add<0x140>(x);  // row_mirror
  return ((read_lane_f64(x, 0) + read_lane_f64(x, 16)) + read_lane_f64(x, 32)) + read_lane_f64(x, 48);
}

__device__ __forceinline__ double np_logaddexp(double x, double y) {  // numpy npy_logaddexp
  if (x == y) return x + 0.693147180559945309417232121458176568;
  double tmp = x - y;
  if (tmp > 0) return x + log1p(exp(-tmp));
  if (tmp <= 0) return y + log1p(exp(tmp));
  return tmp;
}

}  // namespace aehmc
#include "nuts_tree.cuh"  // the scalar state machine of a NUTS transition, shared by all kernel families
namespace aehmc {

// coordinate-wise targets: contribution to U and dU/dq_i
__device__ __forceinline__ void target_elem(const EngineArgs &a, long long i, double q,
                                            double &u, double &g) {
  switch (a.tkind) {
    case AEHMC_T_STD_NORMAL:
      u = 0.5 * (q * q) + AEHMC_LOG_SQRT_2PI;
      g = q;
      break;
    case AEHMC_T_ISO_GAUSSIAN:
      u = q * q;  // U = 0.5 * sum
      g = q;
      break;
#ifdef AEHMC_CUSTOM_TARGET  // run-time compiled copy of these kernels (aehmc_set_custom_target): the user's function
    case AEHMC_T_CUSTOM:
      aehmc_custom_elem(q, i, a.cparams, u, g);
      break;
#endif
    default: {  // AEHMC_T_DIAG_GAUSSIAN
      double s = a.sigma[i];
      double z = (q - a.mu[i]) / s;
      u = 0.5 * (z * z) + a.log_sigma[i] + AEHMC_LOG_SQRT_2PI;
      g = z / s;
    }
  }
}
#ifdef AEHMC_JOINT_TARGET  // run-time compiled copy (aehmc_set_custom_joint_target): the user's joint log-density
// Wave-wide evaluation (D <= 64): lane i holds q_i; returns U = -logp (the same bits in every lane) and this lane's
// dU/dq_i -- forward mode, the derivative seeded at the lane's own coordinate (dual.cuh: JointArg<Dual>).
// A density whose reductions run over far more terms than it has coordinates (a regression's sum over its data rows: the
// tracer sets AEHMC_JOINT_GRAD_SMALL) takes its reverse-mode program here too: the loops spread over the 64 lanes instead
// of every lane running all of them for its own directional derivative.
__device__ __forceinline__ double target_joint(const EngineArgs &a, int lane, int D, double q, double &g) {
#ifdef AEHMC_JOINT_GRAD_SMALL
  __shared__ double aehmc_joint_small[16][128];  // per wavefront: the position row and the gradient row (<= 1024 threads per workgroup)
  double *const qr = aehmc_joint_small[threadIdx.x >> 6], *const gr = qr + 64;
  qr[lane] = lane < D ? q : 0.0;
  gr[lane] = 0.0;
  __threadfence_block();
  const double lp = aehmc_logp_grad(qr, gr, lane, a.cparams);
  __threadfence_block();
  g = -gr[lane];
  __threadfence_block();  // (the rows are rewritten by the next evaluation)
  return -lp;
#else
  const JointArg<Dual> arg{q, lane, D};
  const Dual r = aehmc_logp(arg, a.cparams);
  g = -r.d;
  return -r.v;
#endif
}
#endif
__device__ __forceinline__ double target_finish(const EngineArgs &a, double usum) {
  return (a.tkind == AEHMC_T_ISO_GAUSSIAN || a.tkind == AEHMC_T_DENSE_MVN) ? 0.5 * usum : usum;
}
__device__ __forceinline__ bool target_is_elem(int k) {
  return k == AEHMC_T_STD_NORMAL || k == AEHMC_T_ISO_GAUSSIAN || k == AEHMC_T_DIAG_GAUSSIAN || k == AEHMC_T_CUSTOM;
}
// diagonal / scalar velocity imm o p (metrics.py:47,51,71)
__device__ __forceinline__ double vel_diag(const EngineArgs &a, long long c, long long i, double p) {
  return a.imm[c * a.imm_cs + (a.met_ndim == 0 ? 0 : i)] * p;
}

// One wavefront's pass over the D elements of its chain, four elements per lane in flight:
// `load(i)` returns what element i needs from memory, `use(i, loaded)` computes and stores.
// All loads of a batch are issued before its first store (the pointers in EngineArgs may
// alias as far as the compiler knows, so a plain loop waits for every load before the next
// store and streams at a quarter of HBM speed).  Each lane still visits its elements in
// ascending order: sums are bit-identical to the plain loop.
template <class Load, class Use>
__device__ __forceinline__ void wave_pass(long long D, int lane, Load load, Use use) {
  constexpr int UN = 4;
  for (long long i0 = lane; i0 < D; i0 += 64 * UN) {
    decltype(load(0LL)) vals[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long long i = i0 + 64 * u;
      vals[u] = load(i < D ? i : i0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const long long i = i0 + 64 * u;
      if (i < D) use(i, vals[u]);
    }
  }
}
struct Ld2 { double a, b; };
struct Ld3 { double a, b, c; };
struct Ld4 { double a, b, c, d; };
struct Ld5 { double a, b, c, d, e; };
struct Ld6 { double a, b, c, d, e, f; };
struct Ld9 { double a, b, c, d, e, f, g, h, i; };

// ---------------------------------------------------------------------------------
// Leapfrog stages (integrators.py:54-73).  DO1: p_half = p - (0.5 eps) g.
// DO2: q' = q + (1 eps) v_half, then the target at q' (coordinate-wise targets inline;
// dense target: r = q' - mu is staged for the GEMM).  DO3: p' = p_half - (0.5 eps) g'.
// ---------------------------------------------------------------------------------
// Returns true (and the new potential energy in U_out, on every lane) when the target
// value was completed by this stage set.
template <bool DO1, bool DO2, bool DO3, bool MET_DENSE>
__device__ __forceinline__ bool leap_stages(const EngineArgs &a, long long c, int lane, int dir,
                                            double &U_out) {
  const double step_size = (dir ? 1.0 : -1.0) * (a.eps_c ? a.eps_c[c] : a.eps);
  const double b = 0.5 * step_size, aa = 1 * step_size;
  const size_t row = (size_t)c * a.D;
  const bool elem = target_is_elem(a.tkind);
  const bool tdense = a.tkind == AEHMC_T_DENSE_MVN;
  double usum = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double p = a.cur_p[row + i];
    double gnew = 0.0;
    if (DO1) p = p - b * a.cur_g[row + i];
    if (DO2) {
      double v = MET_DENSE ? a.vhalf[row + i] : vel_diag(a, c, i, p);
      double q = a.cur_q[row + i] + aa * v;
      a.cur_q[row + i] = q;
      if (elem) {
        double u;
        target_elem(a, i, q, u, gnew);
        usum += u;
        a.cur_g[row + i] = gnew;
      } else if (tdense) {
        a.rbuf[row + i] = q - a.mu[i];
      }
    }
    if (DO3) {
      double gi = (DO2 && elem) ? gnew : a.cur_g[row + i];
      if (tdense) usum += a.rbuf[row + i] * gi;
      p = p - b * gi;
    }
    if (DO1 || DO3) a.cur_p[row + i] = p;
  }
  if ((DO2 && elem) || (DO3 && tdense)) {
    U_out = target_finish(a, wave_sum(usum));
    return true;
  }
  return false;
}

// Dense metric, linear mode.  By linearity of v = imm p the two metric products of a
// leapfrog collapse into w' = imm g':  v_half = v - b w,  v' = v_half - b w'.
// PHASE 12: p_half, v_half, q' (+ target / staging of r);  PHASE 3: p', v' (+ U for the
// dense target).  Same return convention as leap_stages.
template <int PHASE>
__device__ __forceinline__ bool leap_linear(const EngineArgs &a, long long c, int lane, int dir,
                                            double &U_out) {
  const double step_size = (dir ? 1.0 : -1.0) * (a.eps_c ? a.eps_c[c] : a.eps);
  const double b = 0.5 * step_size, aa = 1 * step_size;
  const size_t row = (size_t)c * a.D;
  const bool elem = target_is_elem(a.tkind);
  const bool tdense = a.tkind == AEHMC_T_DENSE_MVN;
  double usum = 0.0;
  if (PHASE == 12) {
    wave_pass(a.D, lane,
              [&](long long i) {
                return Ld6{a.cur_p[row + i], a.cur_g[row + i], a.cur_v[row + i], a.cur_w[row + i], a.cur_q[row + i],
                           tdense ? a.mu[i] : 0.0};
              },
              [&](long long i, const Ld6 &x) {
                double p = x.a - b * x.b;
                double v = x.c - b * x.d;
                a.cur_p[row + i] = p;
                a.cur_v[row + i] = v;
                double q = x.e + aa * v;
                a.cur_q[row + i] = q;
                if (elem) {
                  double u, gnew;
                  target_elem(a, i, q, u, gnew);
                  usum += u;
                  a.cur_g[row + i] = gnew;
                } else if (tdense) {
                  a.rbuf[row + i] = q - x.f;
                }
              });
  } else {
    wave_pass(a.D, lane,
              [&](long long i) {
                return Ld5{a.cur_g[row + i], tdense ? a.rbuf[row + i] : 0.0, a.cur_p[row + i], a.cur_v[row + i],
                           a.cur_w[row + i]};
              },
              [&](long long i, const Ld5 &x) {
                if (tdense) usum += x.b * x.a;
                a.cur_p[row + i] = x.c - b * x.a;
                a.cur_v[row + i] = x.d - b * x.e;
              });
  }
  if ((PHASE == 12 && elem) || (PHASE == 3 && tdense)) {
    U_out = target_finish(a, wave_sum(usum));
    return true;
  }
  return false;
}

// The chain's RNG call sites, kept in registers for the duration of a kernel (every lane
// holds the same state); lane 0 writes them back.
struct ChainRng {
  Pcg64 g[4];
};
__device__ __forceinline__ ChainRng rng_load(const EngineArgs &a, long long c) {
  ChainRng r;
#pragma unroll
  for (int k = 0; k < 4; k++) {  // static indices only (a dynamic one sends g[] to scratch)
    r.g[k].state = mk128(0, 0);
    r.g[k].inc = mk128(0, 0);
    if (k < a.nsites) r.g[k] = pcg_load(a.rng + ((size_t)c * a.nsites + k) * 4);
  }
  return r;
}
__device__ __forceinline__ void rng_store(const EngineArgs &a, long long c, int lane, const ChainRng &r,
                                          int first, int last) {
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k >= first && k <= last && k < a.nsites) pcg_store(a.rng + ((size_t)c * a.nsites + k) * 4, r.g[k]);
  }
}

// ---------------------------------------------------------------------------------
// NUTS bookkeeping after one leapfrog of the chain's moving end (cur_*).
// ---------------------------------------------------------------------------------
template <bool MET_DENSE>
__device__ __forceinline__ void copy_cur_to_slot(const EngineArgs &a, size_t row, int lane,
                                                 int slot) {
  for (long long i = lane; i < a.D; i += 64) {
    pick2(a.slot_q, slot)[row + i] = a.cur_q[row + i];
    pick2(a.slot_p, slot)[row + i] = a.cur_p[row + i];
    pick2(a.slot_g, slot)[row + i] = a.cur_g[row + i];
  }
}

template <bool MET_DENSE>
__device__ inline void nuts_begin_expansion(const EngineArgs &a, long long c, int lane,
                                            ChainCtl &ct, int prev_dir, ChainRng &rng) {
  int go_right = rng_bernoulli(rng.g[1], 0.5);  // trajectory.py:516
  ct.dir = go_right;
  ct.step = 0;
  if (prev_dir >= 0 && prev_dir != go_right) {  // cur <- the other end (trajectory.py:518)
    const size_t row = (size_t)c * a.D;
    for (long long i = lane; i < a.D; i += 64) {
      a.cur_q[row + i] = pick2(a.end_q, go_right)[row + i];
      a.cur_p[row + i] = pick2(a.end_p, go_right)[row + i];
      a.cur_g[row + i] = pick2(a.end_g, go_right)[row + i];
      if (MET_DENSE) a.cur_v[row + i] = pick2(a.end_v, go_right)[row + i];
      if (MET_DENSE && a.linear) a.cur_w[row + i] = pick2(a.end_w, go_right)[row + i];
    }
    ct.U_cur = pick2(ct.U_end, go_right);
  }
}

__device__ inline void nuts_write_outputs(const EngineArgs &a, long long c, int lane,
                                          const ChainCtl &ct) {
  const size_t row = (size_t)c * a.D;
  const int s = ct.prop_slot;
  for (long long i = lane; i < a.D; i += 64) {
    a.q[row + i] = pick2(a.slot_q, s)[row + i];
    a.g[row + i] = pick2(a.slot_g, s)[row + i];
    if (a.out.momentum) a.out.momentum[row + i] = pick2(a.slot_p, s)[row + i];
  }
  if (lane == 0) {
    a.U[c] = pick2(ct.U_slot, s);
    a.out.acceptance_probability[c] = ct.acc_prob;
    if (a.out.num_doublings) a.out.num_doublings[c] = ct.ndoubl;
    if (a.out.is_turning) a.out.is_turning[c] = ct.out_turn;
    a.out.is_diverging[c] = ct.out_div;
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = ct.nleap;
  }
}

// expand_once after integrate() returned: trajectory.py:537-608
template <bool MET_DENSE>
__device__ inline void nuts_finalize_expansion(const EngineArgs &a, long long c, int lane,
                                               ChainCtl &ct, bool is_div, bool has_term,
                                               ChainRng &rng) {
  const size_t row = (size_t)c * a.D;
  const int dir = ct.dir, oth = 1 - dir;
  // one pass: moving end <- cur, psum += psub, whole-trajectory U-turn dots (metrics.py:75-104)
  double d_l = 0.0, d_r = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double pc = a.cur_p[row + i], po = pick2(a.end_p, oth)[row + i];
    double vc = MET_DENSE ? a.cur_v[row + i] : vel_diag(a, c, i, pc);
    double vo = MET_DENSE ? pick2(a.end_v, oth)[row + i] : vel_diag(a, c, i, po);
    double s = a.psum[row + i] + a.psub[row + i];
    a.psum[row + i] = s;
    double pl = dir ? po : pc, pr = dir ? pc : po;
    double vl = dir ? vo : vc, vr = dir ? vc : vo;
    double rho = s - (pr + pl) / 2;
    d_l += vl * rho;
    d_r += vr * rho;
    pick2(a.end_q, dir)[row + i] = a.cur_q[row + i];
    pick2(a.end_p, dir)[row + i] = pc;
    pick2(a.end_g, dir)[row + i] = a.cur_g[row + i];
    if (MET_DENSE) pick2(a.end_v, dir)[row + i] = vc;
    if (MET_DENSE && a.linear) pick2(a.end_w, dir)[row + i] = a.cur_w[row + i];
  }
  d_l = wave_sum(d_l);
  d_r = wave_sum(d_r);
  const bool turning = (d_l <= 0) | (d_r <= 0);
  put2(ct.U_end, dir, ct.U_cur);

  // trajectory.py:551-564, proposals.py:105-174 (nuts_tree.cuh)
  if (tree_merge_expansion<true>(ct, is_div, has_term, lane, [&](double pr) { return rng_bernoulli(rng.g[3], pr); }))
    ct.prop_slot ^= 1;
  if (tree_expansion_outcome(ct, is_div, has_term, turning, a.max_exp)) {
    nuts_write_outputs(a, c, lane, ct);
    ct.done = 1;  // caller keeps the chain alive while a phantom scan is pending
  } else {
    ct.j += 1;
    nuts_begin_expansion<MET_DENSE>(a, c, lane, ct, dir, rng);
  }
}

// dynamic_integration.integrate body, one step: trajectory.py:195-305
// FUSE = 1 (dense metric, linear mode): the pass below also performs the last leapfrog stage
// (leap_linear<3>: p' = p_half - b g', v' = v_half - b w', U' for the dense target) on the fly and
// the first U-turn level of an odd step, so p', v' and the running momentum sum are not read back
// (3.5 of the ~22 vectors this kernel moves per chain and step).  FUSE = 2 (diagonal / scalar
// metric, coordinate-wise target): the whole leapfrog (leap_stages<1,1,1>) runs inside that pass --
// one trip to memory per step instead of three dependent ones.  Every lane adds the same terms in
// the same order as the separate passes: identical bits.
template <bool MET_DENSE, int FUSE = 0>
__device__ inline void nuts_book(const EngineArgs &a, long long c, int lane, ChainCtl &ct,
                                 ChainRng &rng) {
  const size_t row = (size_t)c * a.D;
  const int step = ct.step;
  if (!ct.phantom) ct.nleap += 1;
  const TreeIdx ti = tree_step_indices(step, ct.tmin, ct.tmax);  // termination.py:109-113, 192-235 in closed form
  const int tmin = ti.tmin, tmax = ti.tmax;
  const bool even = (step & 1) == 0;
  double *ckp = a.ckp + ((size_t)tmax * a.C + c) * a.D;
  double *cks = a.cks + ((size_t)tmax * a.C + c) * a.D;
  double *ckv = MET_DENSE ? a.ckv + ((size_t)tmax * a.C + c) * a.D : nullptr;
  double kd = 0.0;
  double f_dl = 0.0, f_dr = 0.0;  // FUSE3: U-turn dots of level tmax (odd steps)
  constexpr bool FUSE3 = FUSE == 1;
  const bool f_turn = FUSE != 0 && step >= 1 && tmax >= tmin;
  if (FUSE == 2) {
    const double step_size = (ct.dir ? 1.0 : -1.0) * (a.eps_c ? a.eps_c[c] : a.eps);
    const double b = 0.5 * step_size, aa = 1 * step_size;
    const double *kp = a.ckp + ((size_t)tmax * a.C + c) * a.D;
    const double *ks = a.cks + ((size_t)tmax * a.C + c) * a.D;
    double usum = 0.0;
    wave_pass(a.D, lane,
              [&](long long i) {
                return Ld6{a.cur_p[row + i], a.cur_g[row + i], a.cur_q[row + i], step == 0 ? 0.0 : a.psub[row + i],
                           f_turn ? kp[i] : 0.0, f_turn ? ks[i] : 0.0};
              },
              [&](long long i, const Ld6 &x) {
                double p = x.a - b * x.b;                       // leap_stages<1,1,1>
                const double q = x.c + aa * vel_diag(a, c, i, p);
                a.cur_q[row + i] = q;
                double u, gnew;
                target_elem(a, i, q, u, gnew);
                usum += u;
                a.cur_g[row + i] = gnew;
                p = p - b * gnew;
                a.cur_p[row + i] = p;
                const double v = vel_diag(a, c, i, p);          // bookkeeping pass 1
                kd += v * p;
                const double s2 = (step == 0) ? p : x.d + p;
                a.psub[row + i] = s2;
                if (even) {
                  ckp[i] = p;
                  cks[i] = s2;
                }
                if (f_turn) {                                   // first level of is_iterative_turning
                  const double pl = x.e;
                  const double vl = vel_diag(a, c, i, pl);
                  const double sub = s2 - x.f + pl;
                  const double rho = sub - (p + pl) / 2;
                  f_dl += vl * rho;
                  f_dr += v * rho;
                }
              });
    ct.U_cur = target_finish(a, wave_sum(usum));
  } else if (FUSE3) {
    const double step_size = (ct.dir ? 1.0 : -1.0) * (a.eps_c ? a.eps_c[c] : a.eps);
    const double b = 0.5 * step_size;
    const bool tdense = a.tkind == AEHMC_T_DENSE_MVN;
    const double *kp = a.ckp + ((size_t)tmax * a.C + c) * a.D;
    const double *ks = a.cks + ((size_t)tmax * a.C + c) * a.D;
    const double *kv = a.ckv + ((size_t)tmax * a.C + c) * a.D;
    double usum = 0.0;
    wave_pass(a.D, lane,
              [&](long long i) {
                return Ld9{a.cur_g[row + i], tdense ? a.rbuf[row + i] : 0.0, a.cur_p[row + i], a.cur_v[row + i],
                           a.cur_w[row + i], step == 0 ? 0.0 : a.psub[row + i],
                           f_turn ? kp[i] : 0.0, f_turn ? kv[i] : 0.0, f_turn ? ks[i] : 0.0};
              },
              [&](long long i, const Ld9 &x) {
                if (tdense) usum += x.b * x.a;          // leap_linear<3>
                const double p = x.c - b * x.a;
                const double v = x.d - b * x.e;
                a.cur_p[row + i] = p;
                a.cur_v[row + i] = v;
                kd += v * p;                            // bookkeeping pass 1
                const double s = (step == 0) ? p : x.f + p;
                a.psub[row + i] = s;
                if (even) {
                  ckp[i] = p;
                  cks[i] = s;
                  ckv[i] = v;
                }
                if (f_turn) {                           // first level of is_iterative_turning
                  const double pl = x.g, vl = x.h;
                  const double sub = s - x.i + pl;
                  const double rho = sub - (p + pl) / 2;
                  f_dl += vl * rho;
                  f_dr += v * rho;
                }
              });
    if (tdense) ct.U_cur = target_finish(a, wave_sum(usum));
  } else
  wave_pass(a.D, lane,
            [&](long long i) {
              return Ld3{a.cur_p[row + i], MET_DENSE ? a.cur_v[row + i] : 0.0, step == 0 ? 0.0 : a.psub[row + i]};
            },
            [&](long long i, const Ld3 &x) {
              double p = x.a;
              double v = MET_DENSE ? x.b : vel_diag(a, c, i, p);
              kd += v * p;
              double s = (step == 0) ? p : x.c + p;  // trajectory.py:278,243
              a.psub[row + i] = s;
              if (even) {  // termination.py:115-124
                ckp[i] = p;
                cks[i] = s;
                if (MET_DENSE) ckv[i] = v;
              }
            });
  kd = wave_sum(kd);
  ct.tmin = tmin;
  ct.tmax = tmax;
  // proposals.py:19-62, then progressive_uniform_sampling proposals.py:72-102 (+ :141-144) -- nuts_tree.cuh
  const TreePoint np = tree_new_point(ct.H0, ct.U_cur, kd, a.thr);
  const bool div = np.div;
  bool term = false;
  if (tree_sample_step<true>(ct, step, np, lane, [&](double pr) { return rng_bernoulli(rng.g[2], pr); })) {
    copy_cur_to_slot<MET_DENSE>(a, row, lane, ct.prop_slot ^ 1);  // sub-trajectory proposal <- the new point
    put2(ct.U_slot, ct.prop_slot ^ 1, ct.U_cur);
  }
  if (step >= 1) {
    // is_iterative_turning termination.py:133-187
    if (tmax >= tmin) {
      int idx = tmax;
      bool crit = false;
      for (;;) {
        const double *kp = a.ckp + ((size_t)idx * a.C + c) * a.D;
        const double *ks = a.cks + ((size_t)idx * a.C + c) * a.D;
        const double *kv = MET_DENSE ? a.ckv + ((size_t)idx * a.C + c) * a.D : nullptr;
        double d_l = 0.0, d_r = 0.0;
        if (FUSE != 0 && idx == tmax) {
          d_l = f_dl;
          d_r = f_dr;
        } else
        wave_pass(a.D, lane,
                  [&](long long i) {
                    return Ld6{kp[i], a.cur_p[row + i], MET_DENSE ? kv[i] : 0.0, MET_DENSE ? a.cur_v[row + i] : 0.0,
                               a.psub[row + i], ks[i]};
                  },
                  [&](long long i, const Ld6 &x) {
                    double pl = x.a, pr = x.b;
                    double vl = MET_DENSE ? x.c : vel_diag(a, c, i, pl);
                    double vr = MET_DENSE ? x.d : vel_diag(a, c, i, pr);
                    double sub = x.e - x.f + pl;
                    double rho = sub - (pr + pl) / 2;
                    d_l += vl * rho;
                    d_r += vr * rho;
                  });
        d_l = wave_sum(d_l);
        d_r = wave_sum(d_r);
        crit = (d_l <= 0) | (d_r <= 0);
        bool reached = (idx - 1) < tmin;
        idx -= 1;
        if (crit || reached) break;
      }
      term = crit;
    }
  }
  const TreeControl tc = tree_step_control(ct, step, div, term);
  if (tc.finalize) {
    nuts_finalize_expansion<MET_DENSE>(a, c, lane, ct, tc.fin_div, tc.fin_term, rng);
    if (step == 0) {
      // trajectory.py:336: integrate() returns the first-step tuple, yet the scan still
      // executes (and draws from site #3): finalized now, the chain keeps stepping as a phantom.
      ct.done = 0;
      ct.phantom = 1;
      ct.step = 1;
    }
  }
}

// nuts.py:113-125 after the momentum is in cur_p (and cur_v for a dense metric)
// U_in (optional): the chain's potential energy from a register instead of a.U[c] (kernels that run several
// transitions per launch keep it there)
template <bool MET_DENSE>
__device__ inline void nuts_init_chain(const EngineArgs &a, long long c, int lane, ChainCtl &ct,
                                       ChainRng &rng, const double *U_in = nullptr) {
  const size_t row = (size_t)c * a.D;
  double kd = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double q = a.q[row + i], g = a.g[row + i], p = a.cur_p[row + i];
    double v = MET_DENSE ? a.cur_v[row + i] : vel_diag(a, c, i, p);
    kd += v * p;
    a.cur_q[row + i] = q;
    a.cur_g[row + i] = g;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      a.end_q[e][row + i] = q;
      a.end_p[e][row + i] = p;
      a.end_g[e][row + i] = g;
      if (MET_DENSE) a.end_v[e][row + i] = v;
      if (MET_DENSE && a.linear) a.end_w[e][row + i] = a.cur_w[row + i];
    }
    a.slot_q[0][row + i] = q;
    a.slot_p[0][row + i] = p;
    a.slot_g[0][row + i] = g;
    a.psum[row + i] = p;
  }
  kd = wave_sum(kd);
  const double U = U_in ? *U_in : a.U[c];
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
  nuts_begin_expansion<MET_DENSE>(a, c, lane, ct, -1, rng);
}

// metrics.py:65-68: z ~ N(0, I) from site #1; diagonal metric scales in place,
// dense metric stages z for the GEMM with L^-T.
template <bool MET_DENSE>
__device__ inline void draw_momentum(const EngineArgs &a, long long c, int lane, Pcg64 &g1) {
  const size_t row = (size_t)c * a.D;
  double *dst = MET_DENSE ? a.zbuf : a.cur_p;
  const double *sm = a.sqrt_mass + c * a.imm_cs;
  const bool scalar = a.met_ndim == 0;
  wave_normals(g1, a.D, [=](long long i, double z) {
    dst[row + i] = MET_DENSE ? z : (scalar ? sm[0] : sm[i]) * z;
  });
  __threadfence_block();  // elements were written by arbitrary lanes of this wave
}

// ------------------------------------------------------------------- kernels ------
// the wave index is made explicitly wave-uniform so that per-chain scalars (ChainCtl, RNG
// states) are fetched with scalar loads and live in SGPRs
#define AEHMC_CHAIN_OF_WAVE()                                                   \
  const int lane = threadIdx.x & 63;                                            \
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) +               \
                      __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  \
  if (c >= a.C) return;

template <bool MET_DENSE>
__global__ __launch_bounds__(256) void k_nuts_draw(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainRng rng = rng_load(a, c);
  draw_momentum<MET_DENSE>(a, c, lane, rng.g[0]);
  rng_store(a, c, lane, rng, 0, 0);
}
template <bool MET_DENSE>
__global__ __launch_bounds__(256) void k_nuts_init(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainRng rng = rng_load(a, c);
  ChainCtl ct;
  nuts_init_chain<MET_DENSE>(a, c, lane, ct, rng);
  rng_store(a, c, lane, rng, 1, 1);
  if (lane == 0) a.ctl[c] = ct;
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_nuts_begin_diag(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainRng rng = rng_load(a, c);
  ChainCtl ct;
  draw_momentum<false>(a, c, lane, rng.g[0]);
  nuts_init_chain<false>(a, c, lane, ct, rng);
  rng_store(a, c, lane, rng, 0, 1);
  if (lane == 0) a.ctl[c] = ct;
}
// Whole NUTS transition of a chain in ONE launch (diagonal / scalar metric, coordinate-wise
// target): the wave that owns the chain loops leapfrog + bookkeeping until its tree is
// done; chains of different depth simply retire at different times.  Same device
// functions, hence same bits, as the lock-step path.
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_nuts_fused(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainRng rng = rng_load(a, c);
  ChainCtl ct;
  draw_momentum<false>(a, c, lane, rng.g[0]);
  nuts_init_chain<false>(a, c, lane, ct, rng);
  while (!ct.done) {
    double U_new = 0.0;
    if (leap_stages<true, true, true, false>(a, c, lane, ct.dir, U_new)) ct.U_cur = U_new;
    nuts_book<false>(a, c, lane, ct, rng);
  }
  rng_store(a, c, lane, rng, 0, 3);
}
// ---- small dense problems: the whole transition of a chain in ONE launch, mat-vecs inside the wavefront ----------
// Dense inverse mass matrix (shared or per chain) and / or dense-precision target with D <= 64 -- the classic
// full-mass-matrix use.  On the lock-step path a leapfrog of such a problem is four launches around two D x D "GEMMs"
// and costs ~70 us whatever the chain count (tools/debug/small_dense.py: 1.5 ms per transition at D = 50, 4096
// chains).  Here the wavefront that owns the chain keeps position, momentum, gradient and velocity in registers,
// element i in lane i (NUTS: k_nuts_resident's DENSE instantiations, nuts_resident.cuh; HMC: k_hmc_fused_dense
// below): the matrices sit TRANSPOSED in LDS, lane i forms row i of a product with the operand's elements broadcast
// from registers (v_readlane), k ascending from 0.0 -- the order of the reference's dot products (the restatement's
// too), one rounding per product and per sum.  Literal dense mode (metrics.py:71: imm p_half and imm p' are formed,
// 3 products per leapfrog).
constexpr int FUSED_DENSE_MAX_D = 64;
constexpr int JOINT_ROWS_MAX_D = 2048;  // joint (non-separable) user targets on the lock-step path: four wavefronts' position and gradient rows in 128 KB of LDS (k_target_joint_rows)
constexpr int FUSED_DENSE_BLOCK = 512;  // eight chains per workgroup share the matrices
// y[i] = sum_k M[i][k] x[k] for i < D; MT = M transposed in LDS (MT[k * D + i] = M[i][k]); x, y rows in global memory,
// element i read and written by lane i only
// (eight matrix elements are fetched ahead of the eight FMAs that use them: a few wavefronts per SIMD cannot hide a
// load per dependent FMA; the sum still runs k = 0, 1, 2, ... from 0.0)
// operand and result in registers: lane k < D holds x[k] (other lanes are never read), lane i < D returns y[i]
__device__ __forceinline__ double wave_matvec_reg(const double *MT, double xl, int D, int lane) {
  const double *col = MT + (lane < D ? lane : 0);
  double acc = 0.0;
  int k = 0;
  for (; k + 8 <= D; k += 8) {
    double m[8];
#pragma unroll
    for (int u = 0; u < 8; u++) m[u] = col[(k + u) * D];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += m[u] * read_lane_f64(xl, k + u);
  }
  for (; k < D; k++) acc += col[k * D] * read_lane_f64(xl, k);
  return acc;
}
__device__ __forceinline__ void wave_matvec_lds(const double *MT, const double *x, double *y, int D, int lane) {
  const bool on = lane < D;
  const double acc = wave_matvec_reg(MT, on ? x[lane] : 0.0, D, lane);
  if (on) y[lane] = acc;
}
// One leapfrog (integrators.py:63-100 with metrics.py:71's literal products) of a D <= 64 chain whose position,
// momentum and gradient sit in registers, element i in lane i: the arithmetic of leap_stages<1,1,1> with the
// products above in between, no trip through the work arrays between the stages (a dependent L2 round trip each).
// Returns the new potential energy; lanes >= D carry don't-care values (never read by the products, masked out
// of the energy sum).
template <bool MD, bool TD>
__device__ __forceinline__ double leap_small_dense(const EngineArgs &a, long long c, int lane, double step_size,
                                                   const double *immW, const double *PT, int D, double &q, double &p,
                                                   double &g) {  // step_size: signed (direction * eps)
  const double b = 0.5 * step_size, aa = 1 * step_size;
  const bool on = lane < D;
  const long long il = on ? lane : 0;
  p = p - b * g;
  const double v = MD ? wave_matvec_reg(immW, p, D, lane) : vel_diag(a, c, il, p);
  q = q + aa * v;
  double u;
#ifdef AEHMC_JOINT_TARGET
  if (!TD && a.tkind == AEHMC_T_JOINT) {  // the whole density in one wave-wide evaluation
    const double U = target_joint(a, lane, D, q, g);
    p = p - b * g;
    return U;
  }
#endif
  if (TD) {
    const double r = q - a.mu[il];
    g = wave_matvec_reg(PT, r, D, lane);  // dU/dq = P r
    u = r * g;
  } else {
    target_elem(a, il, q, u, g);
  }
  p = p - b * g;
  return target_finish(a, wave_sum(on ? u : 0.0));
}
// the same product with the chain's OWN row-major matrix in global memory (per-chain dense metrics, what
// is_mass_matrix_full adaptation produces): lane i walks row i, a 8 D byte stride between lanes -- used once per
// transition (the momentum draw's L^-T z); the two products per leapfrog read a transposed copy instead
__device__ __forceinline__ double wave_matvec_rows_reg(const double *M, double xl, int D, int lane) {
  const double *mrow = M + (size_t)(lane < D ? lane : 0) * D;
  double acc = 0.0;
  int k = 0;
  for (; k + 8 <= D; k += 8) {
    double m[8];
#pragma unroll
    for (int u = 0; u < 8; u++) m[u] = mrow[k + u];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += m[u] * read_lane_f64(xl, k + u);
  }
  for (; k < D; k++) acc += mrow[k] * read_lane_f64(xl, k);
  return acc;
}
__device__ __forceinline__ void wave_matvec_rows(const double *M, const double *x, double *y, int D, int lane) {
  const bool on = lane < D;
  const double acc = wave_matvec_rows_reg(M, on ? x[lane] : 0.0, D, lane);
  if (on) y[lane] = acc;
}
// per-chain metric: the wavefront writes the literal transpose of its chain's inverse mass matrix to workspace
// once per transition, so that the 2 products per leapfrog read D contiguous doubles per k (same values, same
// k-ascending sums as the rows; the strided walk above costs 16x the cache lines)
__device__ __forceinline__ void wave_transpose_to(const double *M, double *MT, int D, int lane) {
  for (int e = lane; e < D * D; e += 64) {
    const int i = e / D, k = e % D;
    MT[k * D + i] = M[e];
  }
  __threadfence();  // other lanes of this wavefront read what this lane wrote
}
// MODE: stage set; BOOK: run the NUTS bookkeeping afterwards
template <bool DO1, bool DO2, bool DO3, bool MET_DENSE, bool BOOK>
__global__ __launch_bounds__(256) void k_step(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainCtl ct = a.ctl[c];
  if (ct.done) return;
  double U_new = 0.0;
  bool has_U = false;
  if (DO1 && DO2 && DO3 && !MET_DENSE && BOOK) {  // diagonal metric, coordinate-wise target: one fused pass
    ChainRng rng = rng_load(a, c);
    nuts_book<false, 2>(a, c, lane, ct, rng);
    rng_store(a, c, lane, rng, 1, 3);
    if (lane == 0) a.ctl[c] = ct;
    return;
  }
  if (DO1 || DO2 || DO3) has_U = leap_stages<DO1, DO2, DO3, MET_DENSE>(a, c, lane, ct.dir, U_new);
  if (BOOK) {
    if (has_U) ct.U_cur = U_new;
    ChainRng rng = rng_load(a, c);
    nuts_book<MET_DENSE>(a, c, lane, ct, rng);
    rng_store(a, c, lane, rng, 1, 3);
    if (lane == 0) a.ctl[c] = ct;
  } else if (has_U && lane == 0) {
    a.ctl[c].U_cur = U_new;
  }
}

template <int PHASE, bool BOOK>
__global__ __launch_bounds__(256) void k_step_linear(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainCtl ct = a.ctl[c];
  if (ct.done) return;
  double U_new = 0.0;
  if ((PHASE & 3) == 3 && BOOK) {  // last leapfrog stage fused into the bookkeeping pass
    ChainRng rng = rng_load(a, c);
    nuts_book<true, 1>(a, c, lane, ct, rng);
    rng_store(a, c, lane, rng, 1, 3);
    // PHASE 3 + 12: a chain that goes on takes the first stages of its NEXT leapfrog right here -- the
    // vectors the bookkeeping has just written are still in L2, and the step needs one launch less
    if ((PHASE & 12) == 12 && !ct.done) {
      __threadfence_block();
      double U_next = 0.0;
      if (leap_linear<12>(a, c, lane, ct.dir, U_next)) ct.U_cur = U_next;
    }
    if (lane == 0) a.ctl[c] = ct;
    return;
  }
  bool has_U = leap_linear<PHASE>(a, c, lane, ct.dir, U_new);
  if (BOOK) {
    if (has_U) ct.U_cur = U_new;
    ChainRng rng = rng_load(a, c);
    nuts_book<true>(a, c, lane, ct, rng);
    rng_store(a, c, lane, rng, 1, 3);
    if (lane == 0) a.ctl[c] = ct;
  } else if (has_U && lane == 0) {
    a.ctl[c].U_cur = U_new;
  }
}

// ascending list of live chains + count (one 1024-thread block; deterministic order).  Runs after every lock-step
// of a dense problem: thread t owns a contiguous run of chains, reads their flags ONCE (loads in flight together,
// kept as a bit mask), ranks itself with a shuffle scan inside its wavefront and the 16 wavefront totals.
AEHMC_TU_LOCAL __global__ __launch_bounds__(1024) void k_compact(const ChainCtl *ctl, long long C, int *row_idx,
                                                  int *n_rows, int *host_slot) {
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long long per = (C + 1023) / 1024;
  const long long lo = t * per, hi = (lo + per < C) ? lo + per : C;
  unsigned long long mask = 0;  // flags of the first 64 chains of the run (every chain up to C = 65536)
  int n = 0;
#pragma unroll 8
  for (long long c = lo; c < hi; c++) {
    const int live = ctl[c].done ? 0 : 1;
    n += live;
    if (c - lo < 64) mask |= (unsigned long long)live << (c - lo);
  }
  int incl = n;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int sw = wsum[i];
    base += i < w ? sw : 0;
    total += sw;
  }
  int pos = base + incl - n;
  for (long long c = lo; c < hi; c++) {
    const bool live = (c - lo < 64) ? ((mask >> (c - lo)) & 1) != 0 : !ctl[c].done;
    if (live) row_idx[pos++] = (int)c;
  }
  if (t == 0) {
    *n_rows = total;
    if (host_slot) *host_slot = total;
  }
}

AEHMC_TU_LOCAL __global__ void k_count_active(const ChainCtl *ctl, long long C, int *out) {
  __shared__ int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  int n = 0;
  for (long long c = threadIdx.x; c < C; c += blockDim.x) n += ctl[c].done ? 0 : 1;
  atomicAdd(&s, n);
  __syncthreads();
  if (threadIdx.x == 0) *out = s;
}

// ---- HMC (hmc.py:77-124, 157-204; trajectory.py:31-107) ---------------------------
template <bool MET_DENSE>
__device__ inline ChainCtl hmc_init_chain(const EngineArgs &a, long long c, int lane, const double *U_in = nullptr) {
  const size_t row = (size_t)c * a.D;
  double kd = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double p = a.cur_p[row + i];
    double v = MET_DENSE ? a.cur_v[row + i] : vel_diag(a, c, i, p);
    kd += v * p;
    a.cur_q[row + i] = a.q[row + i];
    a.cur_g[row + i] = a.g[row + i];
    a.slot_p[0][row + i] = p;  // initial momentum, returned on rejection
  }
  kd = wave_sum(kd);
  ChainCtl ct = {};
  ct.U_cur = U_in ? *U_in : a.U[c];
  ct.H0 = ct.U_cur + 0.5 * kd;  // hmc.py:187
  ct.dir = 1;
  if (lane == 0) a.ctl[c] = ct;
  return ct;
}
template <bool MET_DENSE>
__global__ __launch_bounds__(256) void k_hmc_init(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  hmc_init_chain<MET_DENSE>(a, c, lane);
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_hmc_begin_diag(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  ChainRng rng = rng_load(a, c);
  draw_momentum<false>(a, c, lane, rng.g[0]);
  rng_store(a, c, lane, rng, 0, 0);
  hmc_init_chain<false>(a, c, lane);
}
// hmc.py:185-204 after the L leapfrogs: flip, energy difference, accept / reject, outputs.  The accept draw comes
// from `g2` (site #2), which the caller owns (registers, for kernels that run several transitions per launch).
struct HmcEnd {
  int acc, is_div;
  double pa;
};
template <bool MET_DENSE>
__device__ inline HmcEnd hmc_end_chain_rng(const EngineArgs &a, long long c, int lane, const ChainCtl &ct, long long L,
                                           Pcg64 &g2) {
  const size_t row = (size_t)c * a.D;
  double kd = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double p = -1.0 * a.cur_p[row + i];  // hmc.py:185 momentum flip
    double v = MET_DENSE ? -1.0 * a.cur_v[row + i] : vel_diag(a, c, i, p);
    kd += v * p;
  }
  kd = wave_sum(kd);
  double new_energy = ct.U_cur + 0.5 * kd;
  double delta = ct.H0 - new_energy;
  if (isnan(delta)) delta = -INFINITY;
  int is_div = fabs(delta) > a.thr;
  double pa = exp(delta);
  if (pa > 1.0) pa = 1.0;
  if (pa < 0.0) pa = 0.0;
  int acc = rng_bernoulli(g2, pa);  // hmc.py:193-194
  for (long long i = lane; i < a.D; i += 64) {
    if (acc) {
      a.q[row + i] = a.cur_q[row + i];
      a.g[row + i] = a.cur_g[row + i];
      if (a.out.momentum) a.out.momentum[row + i] = -1.0 * a.cur_p[row + i];
    } else if (a.out.momentum) {
      a.out.momentum[row + i] = a.slot_p[0][row + i];
    }
  }
  if (lane == 0) {
    if (acc) a.U[c] = ct.U_cur;
    a.out.acceptance_probability[c] = pa;
    a.out.is_diverging[c] = is_div;
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = L;
    if (a.out.is_turning) a.out.is_turning[c] = acc;  // HMC: reused as the accept flag
  }
  HmcEnd e = {acc, is_div, pa};
  return e;
}
template <bool MET_DENSE>
__device__ inline void hmc_end_chain(const EngineArgs &a, long long c, int lane, const ChainCtl &ct, long long L) {
  Pcg64 g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
  hmc_end_chain_rng<MET_DENSE>(a, c, lane, ct, L, g2);
  if (lane == 0) pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, g2);
}
template <bool MET_DENSE>
__global__ __launch_bounds__(256) void k_hmc_end(EngineArgs a, long long L) {
  AEHMC_CHAIN_OF_WAVE();
  const ChainCtl ct = a.ctl[c];
  hmc_end_chain<MET_DENSE>(a, c, lane, ct, L);
}
// The HMC transition of a small dense problem in one launch (see above: same matrices in LDS, same
// in-wavefront products, literal dense mode); the chain's scalars stay in registers between the stages.
template <bool MD, bool TD, bool PC = false>
__global__ __launch_bounds__(FUSED_DENSE_BLOCK) AEHMC_USER_DENSITY_512 void k_hmc_fused_dense(EngineArgs a, const double *prec, double *imm_ws, long long L,
                                                                         long long nt, double *samples, double *acc_hist,
                                                                         int *div_hist) {
  extern __shared__ __attribute__((aligned(16))) double fd_lds[];
  const int D = (int)a.D, DD = D * D;
  constexpr bool MLDS = MD && !PC;
  double *const immT = fd_lds, *const smT = fd_lds + (MLDS ? DD : 0), *const PT = fd_lds + (MLDS ? 2 * DD : 0);
  for (int e = threadIdx.x; e < DD; e += FUSED_DENSE_BLOCK) {
    const int i = e / D, k = e % D;
    if (MLDS) {
      immT[k * D + i] = a.imm[e];
      smT[k * D + i] = a.sqrt_mass[e];
    }
    if (TD) PT[k * D + i] = prec[e];
  }
  __syncthreads();
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  const double *const immW = PC ? imm_ws + (size_t)c * DD : immT;  // PC: this chain's transposed copy (global)
  if (PC) wave_transpose_to(a.imm + (size_t)c * DD, imm_ws + (size_t)c * DD, D, lane);
  // nt consecutive transitions of the chain in this launch (kernel.sample(nt)); per-transition records optional
  for (long long tt = 0; tt < nt; tt++) {
  {
    Pcg64 g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4);
    draw_momentum<MD>(a, c, lane, g1);
    if (lane == 0) pcg_store(a.rng + (size_t)c * a.nsites * 4, g1);
  }
  if (MD) {
    if (PC) wave_matvec_rows(a.sqrt_mass + (size_t)c * DD, a.zbuf + row, a.cur_p + row, D, lane);
    else wave_matvec_lds(smT, a.zbuf + row, a.cur_p + row, D, lane);
    wave_matvec_lds(immW, a.cur_p + row, a.cur_v + row, D, lane);
  }
  ChainCtl ct = hmc_init_chain<MD>(a, c, lane);
  const bool on = lane < D;
  const size_t el = row + (on ? lane : 0);
  double q = a.cur_q[el], p = a.cur_p[el], g = a.cur_g[el];
  const double eps_c = 1.0 * (a.eps_c ? a.eps_c[c] : a.eps);
  for (long long l = 0; l < L; l++)  // trajectory.py:86-95: the whole trajectory in registers
    ct.U_cur = leap_small_dense<MD, TD>(a, c, lane, eps_c, immW, PT, D, q, p, g);
  if (on) {
    a.cur_q[el] = q;
    a.cur_p[el] = p;
    a.cur_g[el] = g;
  }
  if (MD && L > 0) {
    const double v = wave_matvec_reg(immW, p, D, lane);  // for the final kinetic energy
    if (on) a.cur_v[el] = v;
  }
  hmc_end_chain<MD>(a, c, lane, ct, L);
  if (samples && on) samples[((size_t)tt * a.C + c) * a.D + lane] = a.q[el];  // (what this lane holds now)
  if (lane == 0) {
    if (acc_hist) acc_hist[(size_t)tt * a.C + c] = a.out.acceptance_probability[c];
    if (div_hist) div_hist[(size_t)tt * a.C + c] = a.out.is_diverging[c];
  }
  __threadfence_block();  // the next transition's lanes read the energy lane 0 has just written
  }
}

// ---- new_state / stand-alone building blocks --------------------------------------
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_new_state_elem(EngineArgs a) {
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  double usum = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double u, g;
    target_elem(a, i, a.q[row + i], u, g);
    usum += u;
    a.g[row + i] = g;
  }
  usum = wave_sum(usum);
  if (lane == 0) a.U[c] = target_finish(a, usum);
}
#ifdef AEHMC_JOINT_TARGET
__global__ __launch_bounds__(256) void k_new_state_joint(EngineArgs a) {  // hmc.py:16-40 for a joint target
  AEHMC_CHAIN_OF_WAVE();
  const int D = (int)a.D;
  const bool on = lane < D;
  double g = 0.0;
  const double U = target_joint(a, lane, D, on ? a.q[c * a.D + lane] : 0.0, g);
  if (on) a.g[c * a.D + lane] = g;
  if (lane == 0) a.U[c] = U;
}
// A joint target of any size on the lock-step path (round 5): U and dU/dq of every (live) chain from its position row,
// between the stage kernels -- the place the dense-precision, regression and row-reduction targets are evaluated at
// (engine.hip: launch_leapfrog's target_ext).  One wavefront per chain; the row waits in LDS and the density is evaluated
// ceil(D / 64) times, lane l carrying the derivative with respect to coordinate l + 64 k in pass k (dual.cuh: JointRow):
// O(D^2 / 64) density terms per gradient and wavefront, every lane ends each pass with the same value bits.
// U -> ctl[c].U_cur (leapfrog; finished chains are skipped) or U[c] (new_state).
// A density that comes with its reverse-mode program (AEHMC_JOINT_GRAD: aehmc_logp_grad, emitted by aehmc_amd/tracing.py for
// a traced Python logprob_fn -- round 6; the reference differentiates in reverse mode too, aesara.grad, hmc.py:33-34) takes
// ONE sweep whatever D: the wavefront runs the forward and the adjoint program together, loops over the coordinates
// distributed over its lanes, the gradient accumulated in a second LDS row.
// (device part: `qr` = this wavefront's 2 D doubles of LDS -- the position row and the gradient row; returns U on every lane)
__device__ inline double joint_rows_eval(const EngineArgs &a, const double *q, double *g, double *qr, int lane) {
  const int D = (int)a.D;
#ifdef AEHMC_JOINT_GRAD
  double *const gr = qr + D;
  for (int i = lane; i < D; i += 64) {
    qr[i] = q[i];
    gr[i] = 0.0;
  }
  __threadfence_block();  // (the rows are read and updated through other lanes' addresses)
  const double lp = aehmc_logp_grad(qr, gr, lane, a.cparams);
  __threadfence_block();
  for (int i = lane; i < D; i += 64) g[i] = -gr[i];
  __threadfence_block();  // (the stage that follows reads g through other lanes' addresses)
  return -lp;
#else
  for (int i = lane; i < D; i += 64) qr[i] = q[i];
  __threadfence_block();  // (the row is read back through other lanes' addresses)
  double Uv = 0.0;
  for (int k = 0; k * 64 < D; k++) {
    const JointRow<Dual> arg{qr, lane + 64 * k, D};
    const Dual r = aehmc_logp(arg, a.cparams);
    if (lane + 64 * k < D) g[lane + 64 * k] = -r.d;
    Uv = -r.v;
  }
  __threadfence_block();  // (the stage that follows reads g through other lanes' addresses)
  return Uv;
#endif
}
__global__ __launch_bounds__(256) void k_target_joint_rows(EngineArgs a, const double *q, double *g, double *U, int to_ctl,
                                                           const int *row_idx, const int *n_rows) {
  extern __shared__ __attribute__((aligned(16))) double joint_rows[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  long long c = w;
  if (row_idx) {
    if (w >= *n_rows) return;
    c = row_idx[w];
  } else if (w >= a.C) {
    return;
  }
  if (to_ctl && a.ctl[c].done) return;
  const size_t row = (size_t)c * a.D;
  const double Uv = joint_rows_eval(a, q + row, g + row, joint_rows + (size_t)wave * 2 * a.D, lane);
  if (lane == 0) {
    if (to_ctl) a.ctl[c].U_cur = Uv;
    else U[c] = Uv;
  }
}
// The lock-step loop of a joint target with a scalar / diagonal metric for ONE chain per wavefront, in one launch
// (round 5; what k_nuts_fused is for coordinate-wise targets and k_nuts_pc_dense for per-chain dense metrics): the same
// device functions in the same order -- first stages | density | last stage + bookkeeping -- hence the same bits as
// the lock-step path, without its three launches and its host poll per leapfrog.  m.T transitions per launch.
__global__ __launch_bounds__(256) void k_nuts_joint_rows(EngineArgs a, NutsSampleArgs m) {
  extern __shared__ __attribute__((aligned(16))) double joint_rows[];
  AEHMC_CHAIN_OF_WAVE();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *const qr = joint_rows + (size_t)wave * 2 * a.D;
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
      ct.U_cur = joint_rows_eval(a, a.cur_q + row, a.cur_g + row, qr, lane);
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
// HMC: nt transitions x L leapfrogs in one launch (hmc_run's lock-step loop for one chain per wavefront)
__global__ __launch_bounds__(256) void k_hmc_joint_rows(EngineArgs a, long long L, long long nt, double *samples,
                                                        double *acc_hist, int *div_hist) {
  extern __shared__ __attribute__((aligned(16))) double joint_rows[];
  AEHMC_CHAIN_OF_WAVE();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *const qr = joint_rows + (size_t)wave * 2 * a.D;
  const size_t row = (size_t)c * a.D;
  Pcg64 g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4), g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
  double U_state = a.U[c];
  for (long long tt = 0; tt < nt; tt++) {
    draw_momentum<false>(a, c, lane, g1);
    ChainCtl ct = hmc_init_chain<false>(a, c, lane, &U_state);
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95
      double U_new = 0.0;
      leap_stages<true, true, false, false>(a, c, lane, 1, U_new);
      ct.U_cur = joint_rows_eval(a, a.cur_q + row, a.cur_g + row, qr, lane);
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
#ifdef AEHMC_JOINT_GRAD
// ---- a WORKGROUP per chain (round 6): a traced density whose reductions sweep long data (a regression over 10^5 rows) and
// a call with fewer chains than the GPU has SIMDs.  With a wavefront per chain 1024 chains are one wavefront per SIMD,
// each walking all the rows (latency-bound: profiles/r6/INDEX.md); here W wavefronts run the generated program together
// (aehmc_logp_grad_t<W>: loops over 64 W lanes, sums through LDS) while wavefront 0 runs the lock-step engine's stage /
// bookkeeping functions of the chain between the evaluations.  Same functions in the same order as k_nuts_joint_rows;
// the density's sums are associated differently (64 W partial sums), results agree to rounding.
template <int W>
__device__ inline double joint_wg_eval(const EngineArgs &a, const double *q, double *g, double *qr, int tid) {
  const int D = (int)a.D;
  double *const gr = qr + D;
  for (int i = tid; i < D; i += 64 * W) {
    qr[i] = q[i];
    gr[i] = 0.0;
  }
  __syncthreads();
  const double lp = aehmc_logp_grad_t<W>(qr, gr, tid, a.cparams);
  __syncthreads();
  for (int i = tid; i < D; i += 64 * W) g[i] = -gr[i];
  __threadfence_block();  // (wavefront 0 reads the row behind the caller's barrier)
  return -lp;
}
template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(wg_min_waves(W)))) void k_nuts_joint_wg(EngineArgs a, NutsSampleArgs m) {
  extern __shared__ __attribute__((aligned(16))) double joint_rows[];
  __shared__ int wg_done;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool leader = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const long long c = blockIdx.x;
  double *const qr = joint_rows;
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
      const double Uv = joint_wg_eval<W>(a, a.cur_q + row, a.cur_g + row, qr, tid);
      __syncthreads();
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
template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(wg_min_waves(W)))) void k_hmc_joint_wg(EngineArgs a, long long L, long long nt, double *samples,
                                                         double *acc_hist, int *div_hist) {
  extern __shared__ __attribute__((aligned(16))) double joint_rows[];
  const int tid = threadIdx.x, lane = tid & 63;
  const bool leader = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const long long c = blockIdx.x;
  double *const qr = joint_rows;
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
      const double Uv = joint_wg_eval<W>(a, a.cur_q + row, a.cur_g + row, qr, tid);
      __syncthreads();
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
#endif  // AEHMC_JOINT_GRAD
#endif
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_residual(EngineArgs a, const double *q, double *r) {
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  for (long long i = lane; i < a.D; i += 64) r[row + i] = q[row + i] - a.mu[i];
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_half_dot(EngineArgs a, const double *x, const double *y,
                                                  double *out) {
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  double s = 0.0;
  for (long long i = lane; i < a.D; i += 64) s += x[row + i] * y[row + i];
  s = wave_sum(s);
  if (lane == 0) out[c] = 0.5 * s;
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_vel_diag(EngineArgs a, const double *p, double *v) {
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  for (long long i = lane; i < a.D; i += 64) v[row + i] = vel_diag(a, c, i, p[row + i]);
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_is_turning(EngineArgs a, const double *pl, const double *pr,
                                                    const double *ps, const double *vl,
                                                    const double *vr, int32_t *out) {
  AEHMC_CHAIN_OF_WAVE();
  const size_t row = (size_t)c * a.D;
  double d_l = 0.0, d_r = 0.0;
  for (long long i = lane; i < a.D; i += 64) {
    double rho = ps[row + i] - (pr[row + i] + pl[row + i]) / 2;
    d_l += vl[row + i] * rho;
    d_r += vr[row + i] * rho;
  }
  d_l = wave_sum(d_l);
  d_r = wave_sum(d_r);
  if (lane == 0) out[c] = (d_l <= 0) | (d_r <= 0);
}
// examples/LinearRegression.ipynb:126-166 target, q = [w, log n]: U and dU/dq need sums
// over the N data rows.  One 256-thread block serves 8 chains: every thread streams its
// rows of (X, y) once (from L2) and accumulates sum(x r) and sum(r^2), r = y - x w, for
// all 8 chains; wave shuffles + a fixed-order LDS pass finish the reduction
// (deterministic).  to_ctl: write U into ctl[c].U_cur (leapfrog) or into U[c] (new_state).
constexpr int LINREG_CPB = 8;
// stage 1: workgroup (g, s) sums row slice s for the 8 chains of group g -> part[g][s][16]
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_target_linreg(EngineArgs a, const double *q, double *part,
                                                       int S, int to_ctl) {
  __shared__ double red[4][2 * LINREG_CPB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long grp = blockIdx.x / S;
  const int s = blockIdx.x % S;
  const long long c0 = grp * LINREG_CPB;
  double w[LINREG_CPB], sxr[LINREG_CPB], srr[LINREG_CPB];
  bool any = false;
#pragma unroll
  for (int k = 0; k < LINREG_CPB; k++) {
    const long long c = c0 + k;
    const bool live = c < a.C && !(to_ctl && a.ctl[c].done);
    w[k] = live ? q[c * 2] : 0.0;
    sxr[k] = srr[k] = 0.0;
    any |= live;
  }
  if (!any) return;
  const long long per = (a.N + S - 1) / S;
  const long long lo = s * per, hi = (lo + per < a.N) ? lo + per : a.N;
  // 4 rows per thread and iteration in flight; each thread adds its rows in ascending order
  long long i = lo + tid;
  for (; i + 768 < hi; i += 1024) {
    double xs[4], ys[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      xs[u] = a.X[i + 256 * u];
      ys[u] = a.y[i + 256 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int k = 0; k < LINREG_CPB; k++) lr_term(xs[u], ys[u], w[k], sxr[k], srr[k]);
  }
  for (; i < hi; i += 256) {
    const double x = a.X[i], yy = a.y[i];
#pragma unroll
    for (int k = 0; k < LINREG_CPB; k++) lr_term(x, yy, w[k], sxr[k], srr[k]);
  }
#pragma unroll
  for (int k = 0; k < LINREG_CPB; k++) {
    sxr[k] = wave_sum(sxr[k]);
    srr[k] = wave_sum(srr[k]);
    if (lane == 0) {
      red[wave][2 * k] = sxr[k];
      red[wave][2 * k + 1] = srr[k];
    }
  }
  __syncthreads();
  if (tid < 2 * LINREG_CPB)
    part[((size_t)grp * S + s) * (2 * LINREG_CPB) + tid] =
        ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
// stage 2: one thread per chain adds the S slice sums in order and forms U and dU/dq
// (examples/LinearRegression.ipynb:126-166, q = [w, log n]).  to_ctl: U -> ctl[c].U_cur.
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_linreg_finish(EngineArgs a, const double *q, double *g, double *U,
                                                       const double *part, int S, int to_ctl) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C || (to_ctl && a.ctl[c].done)) return;
  const long long grp = c / LINREG_CPB;
  const int k = (int)(c % LINREG_CPB);
  double s_xr = 0.0, s_rr = 0.0;
  for (int s = 0; s < S; s++) {
    s_xr += part[((size_t)grp * S + s) * (2 * LINREG_CPB) + 2 * k];
    s_rr += part[((size_t)grp * S + s) * (2 * LINREG_CPB) + 2 * k + 1];
  }
  const double ww = q[c * 2], ell = q[c * 2 + 1], n = exp(ell), n2 = n * n, N = (double)a.N;
  const double lp_w = -0.5 * ww * ww - AEHMC_LOG_SQRT_2PI;
  const double lp_n = log(n) - n + ell;
  const double lp_y = -0.5 * (s_rr / n2) - N * AEHMC_LOG_SQRT_2PI - N * ell;
  g[c * 2] = -(-ww + s_xr / n2);
  g[c * 2 + 1] = -(2.0 - n - N + s_rr / n2);
  const double Uv = -(lp_w + lp_n + lp_y);
  if (to_ctl) a.ctl[c].U_cur = Uv;
  else U[c] = Uv;
}

// leapfrog-only driver state: ctl.dir = 1, U in ctl
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_ctl_set(EngineArgs a, const double *U) {
  AEHMC_CHAIN_OF_WAVE();
  if (lane == 0) {
    ChainCtl ct = {};
    ct.dir = 1;
    ct.U_cur = U[c];
    a.ctl[c] = ct;
  }
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_ctl_get_U(EngineArgs a, double *U) {
  AEHMC_CHAIN_OF_WAVE();
  if (lane == 0) U[c] = a.ctl[c].U_cur;
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_rng_normals(uint64_t *rng, long long C, long long n,
                                                     double *out) {
  __shared__ double ztab[ZIG_LDS_DOUBLES];
  const ZigTabLds tab = zig_tab_to_lds(ztab);
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  Pcg64 g = pcg_load(rng + c * 4);
  wave_normals(g, n, [=](long long i, double z) { out[c * n + i] = z; }, tab);
  if ((threadIdx.x & 63) == 0) pcg_store(rng + c * 4, g);
}
// Momentum draw of site #1 for every chain with one wavefront per chain (metrics.py:65-68,
// scalar / diagonal metric): zbuf[c, i] = sqrt_mass[i] * z_i.  The workgroup-per-chain resident
// kernels run this as a pre-pass -- inside them a single wavefront per CU would walk the stream
// while the other waves of the workgroup wait.
// Rows of zbuf are `ld` >= D apart; [D, ld) is filled with zeros.  nt > 1: the momenta of nt consecutive
// transitions (the stream of site #1 serves nothing else), transition tt in zbuf[tt][C][ld].
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_draw_momentum(uint64_t *rng, int nsites, long long C, long long D,
                                                       const double *sqrt_mass, long long sm_cs, int met_ndim,
                                                       double *zbuf, long long ld, int nt) {
  __shared__ double ztab[ZIG_LDS_DOUBLES];
  const ZigTabLds tab = zig_tab_to_lds(ztab);
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  uint64_t *gs = rng + (size_t)c * nsites * 4;
  Pcg64 g = pcg_load(gs);
  const double *sm = sqrt_mass + (size_t)c * sm_cs;
  const bool scalar = met_ndim == 0;
  const PcgLaneJump jump = pcg_lane_jump(g);
  for (int tt = 0; tt < nt; tt++) {
    double *dst = zbuf + ((size_t)tt * C + c) * ld;
    wave_normals(g, D, [=](long long i, double z) { dst[i] = (scalar ? sm[0] : sm[i]) * z; }, tab, jump);
    for (long long i = D + (threadIdx.x & 63); i < ld; i += 64) dst[i] = 0.0;
  }
  if ((threadIdx.x & 63) == 0) pcg_store(gs, g);
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_rng_bernoulli(uint64_t *rng, long long C, long long n,
                                                       const double *p, int32_t *out) {
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  Pcg64 g = pcg_load(rng + c * 4);
  for (long long i = 0; i < n; i++) {
    int b = rng_bernoulli(g, p[c * n + i]);
    if ((threadIdx.x & 63) == 0) out[c * n + i] = b;
  }
  if ((threadIdx.x & 63) == 0) pcg_store(rng + c * 4, g);
}
// ---- per-chain dense metric (what is_mass_matrix_full window adaptation produces) --------
// Every chain owns a D x D inverse mass matrix (C D^2 doubles -- the reference's per-chain
// semantics, mass_matrix.py:12-120, which memory bounds long before AEHMC_PC_DENSE_MAX_D), so the
// metric products are per-chain mat-vecs instead of one GEMM over all chains.  Up to
// AEHMC_PC_LDS_MAX_D the factorisation of a chain's matrix runs in LDS, above it in global memory.
constexpr int AEHMC_PC_LDS_MAX_D = 64;
constexpr int AEHMC_PC_DENSE_MAX_D = 2048;  // (one wavefront factors a chain's matrix; tested up to D = 1024: tests/test_gpu_adaptation.py)
// out[c, i] = sum_j mats[c, i, j] x[c, j]  (j ascending); one wavefront per (live) chain
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_matvec_pc(const double *mats, const double *x, double *out, long long C,
                                                   long long D, const int *row_idx, const int *n_rows) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long long c = w;
  if (row_idx) {
    if (w >= *n_rows) return;
    c = row_idx[w];
  } else if (w >= C) {
    return;
  }
  const double *m = mats + (size_t)c * D * D, *xr = x + (size_t)c * D;
  for (long long i = lane; i < D; i += 64) {
    double s = 0.0;
    for (long long j = 0; j < D; j++) s += m[i * D + j] * xr[j];
    out[(size_t)c * D + i] = s;
  }
}
// The same product for D > 64: workgroup (x, w) forms 64 rows of chain w's product, one row per
// wave at a time with the 64 lanes striding over the columns (coalesced reads of the matrix row)
// and a wave sum per row.
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_matvec_pc_rows(const double *mats, const double *x, double *out,
                                                        long long C, long long D, const int *row_idx,
                                                        const int *n_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long nrb = (D + 63) / 64;  // row blocks per chain; grid = nrb x C in one dimension (no 65535 limit on C)
  const long long w = blockIdx.x / nrb, rb = blockIdx.x % nrb;
  long long c = w;
  if (row_idx) {
    if (w >= *n_rows) return;
    c = row_idx[w];
  }
  const double *m = mats + (size_t)c * D * D, *xr = x + (size_t)c * D;
  for (int k = 0; k < 16; k++) {
    const long long i = rb * 64 + wave * 16 + k;
    if (i >= D) break;
    double s = 0.0;
    for (long long j = lane; j < D; j += 64) s += m[i * D + j] * xr[j];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)c * D + i] = s;
  }
}
// L = chol(A) (lower) and S = L^-T for one D x D matrix worked on by one wavefront (metrics.py:56-58);
// A and S in LDS or in global memory.  `A` is overwritten by L; returns false if A is not positive
// definite.
__device__ inline bool wave_chol_inv_t(double *A, double *S, int D, int lane) {
  bool ok = true;
  for (int k = 0; k < D; k++) {  // right-looking Cholesky, column k
    const double akk = A[k * D + k];
    if (!(akk > 0.0)) ok = false;
    const double lkk = sqrt(akk);
    for (int i = k + lane; i < D; i += 64) A[i * D + k] = (i == k) ? lkk : A[i * D + k] / lkk;
    __threadfence_block();  // (one wavefront; LDS or global)
    for (int idx = lane; idx < (D - k - 1) * (D - k - 1); idx += 64) {
      const int i = k + 1 + idx / (D - k - 1), j = k + 1 + idx % (D - k - 1);
      if (j <= i) A[i * D + j] = A[i * D + j] - A[i * D + k] * A[j * D + k];
    }
    __threadfence_block();  // (one wavefront; LDS or global)
  }
  // X = L^-1 by forward substitution, one column per lane; S = X^T
  for (int col = lane; col < D; col += 64) {
    for (int i = 0; i < D; i++) {
      double v = (i == col) ? 1.0 : 0.0;
      for (int j = col; j < i; j++) v = v - A[i * D + j] * S[col * D + j];  // S[col][j] = X[j][col]
      S[col * D + i] = (i < col) ? 0.0 : v / A[i * D + i];
    }
  }
  __threadfence_block();  // (one wavefront; LDS or global)
  return ok;
}
// sqrt_mass[c] = chol(imm[c])^-T for every chain; *err = 1 if a matrix is not positive definite
AEHMC_TU_LOCAL __global__ __launch_bounds__(64) void k_chol_inv_pc(const double *imm, double *sqrt_mass, long long C, int D,
                                                     int *err, double *work) {
  extern __shared__ __attribute__((aligned(16))) double pc_lds[];  // D <= 64: A [D*D], S [D*D]
  const int lane = threadIdx.x;
  const long long c = blockIdx.x;
  const bool in_lds = D <= AEHMC_PC_LDS_MAX_D;
  double *A = in_lds ? pc_lds : work + (size_t)c * D * D;
  double *S = in_lds ? pc_lds + D * D : sqrt_mass + (size_t)c * D * D;
  for (int i = lane; i < D * D; i += 64) A[i] = imm[(size_t)c * D * D + i];
  __threadfence_block();
  const bool ok = wave_chol_inv_t(A, S, D, lane);
  // S holds X^T laid out as S[col][i] = X[i][col] = (L^-1)[i][col] = (L^-T)[col][i]: row-major L^-T
  if (in_lds)
    for (int i = lane; i < D * D; i += 64) sqrt_mass[(size_t)c * D * D + i] = S[i];
  if (!ok && lane == 0) *err = 1;
}

// ---- window adaptation (window_adaptation.py:119-227), one wave per chain ---------
struct AdaptArgs {
  long long C, D;
  int stage, window_end, last;      // schedule entry of this warm-up step
  double target, gamma, t0, kappa;  // step_size.py:9-14 defaults 0.8, 0.05, 10, 0.75
  const double *p_accept, *position;
  aehmc_adapt_state s;
};
// The per-chain arithmetic of one warm-up update, shared by k_adapt_update and the kernels that run the
// whole warm-up in one launch (nuts_linreg.cuh) -- the same instruction sequence, hence the same bits.
struct DualAvg {  // algorithms.py:9-14
  long long step;
  double x, x_avg, g_avg, mu;
};
// algorithms.py:104-115 + step_size.py:97-98: returns the next step size exp(x)
__device__ __forceinline__ double adapt_da_update(DualAvg &d, double target, double p_accept, double gamma, double t0,
                                                  double kappa) {
  const double x_old = d.x;
  const double eta = 1.0 / ((double)d.step + t0);
  const double gradient = target - p_accept;
  d.g_avg = (1.0 - eta) * d.g_avg + eta * gradient;
  d.x = d.mu - (sqrt((double)d.step) / gamma) * d.g_avg;
  const double x_eta = pow((double)d.step, -kappa);
  d.x_avg = x_eta * x_old + (1.0 - x_eta) * d.x_avg;
  d.step += 1;
  return exp(d.x);
}
// window_adaptation.py:177-178: dual averaging restarted around the current step size
__device__ __forceinline__ void adapt_da_restart(DualAvg &d, double step_size) {
  d.mu = step_size;  // da_init(step_size): the step size itself, not its log
  d.step = 1;
  d.x = 0.0;
  d.x_avg = 0.0;
  d.g_avg = 0.0;
}
// algorithms.py:187-197, one coordinate (n already counts the new draw)
__device__ __forceinline__ void adapt_welford_elem(double v, long long n, double &mean, double &m2) {
  const double delta = v - mean;
  mean = mean + delta / (double)n;
  const double ud = v - mean;
  m2 = m2 + ud * delta;
}
// mass_matrix.py:83-118 (diagonal) + metrics.py:45,49, one coordinate; the Welford state is reset
__device__ __forceinline__ void adapt_window_end_elem(long long n, double &mean, double &m2, double &imm,
                                                      double &sqrt_mass) {
  const double nn = (double)n;
  const double cov = m2 / (double)(n - 1);
  imm = (nn / (nn + 5)) * cov + 1e-3 * (5 / (nn + 5));
  sqrt_mass = sqrt(1.0 / imm);
  mean = 0.0;
  m2 = 0.0;
}

AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_adapt_init(AdaptArgs a, double initial_step_size) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= a.C) return;
  if (a.s.full) {  // mass_matrix.py:56-57: identity
    for (long long i = lane; i < a.D; i += 64) a.s.wc_mean[c * a.D + i] = 0.0;
    for (long long i = lane; i < a.D * a.D; i += 64) {
      const double e = (i / a.D == i % a.D) ? 1.0 : 0.0;
      a.s.wc_m2[c * a.D * a.D + i] = 0.0;
      a.s.imm[c * a.D * a.D + i] = e;
      a.s.sqrt_mass[c * a.D * a.D + i] = e;
    }
  } else
  for (long long i = lane; i < a.D; i += 64) {  // mass_matrix.py:37-61
    a.s.wc_mean[c * a.D + i] = 0.0;
    a.s.wc_m2[c * a.D + i] = 0.0;
    a.s.imm[c * a.D + i] = 1.0;
    a.s.sqrt_mass[c * a.D + i] = 1.0;
  }
  if (lane == 0) {  // algorithms.py:56-76, window_adaptation.py:139-140
    a.s.da_step[c] = 1;
    a.s.da_x[c] = 0.0;
    a.s.da_x_avg[c] = 0.0;
    a.s.da_g_avg[c] = 0.0;
    a.s.da_mu[c] = initial_step_size;
    a.s.wc_n[c] = 0;
    a.s.step_size[c] = exp(0.0);
  }
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_adapt_update(AdaptArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= a.C) return;
  // dual averaging, fast and slow stages alike (algorithms.py:104-115, step_size.py:97-98)
  DualAvg da = {a.s.da_step[c], a.s.da_x[c], a.s.da_x_avg[c], a.s.da_g_avg[c], a.s.da_mu[c]};
  double step_size = adapt_da_update(da, a.target, a.p_accept[c], a.gamma, a.t0, a.kappa);
  long long n = a.s.wc_n[c];
  if (a.s.full) {
    // full covariance (algorithms.py:187-197 with np.outer, mass_matrix.py:83-118), one wavefront per
    // workgroup: delta / updated delta of the whole position sit in LDS, the D x D arrays in HBM; the
    // window-end factorisation runs in LDS (D <= 64) or in a.s.work / sqrt_mass
    extern __shared__ __attribute__((aligned(16))) double ad_lds[];  // delta [D], ud [D], then A, S [D*D] (D <= 64)
    double *const fl_delta = ad_lds, *const fl_ud = ad_lds + a.D;
    const long long DD = a.D * a.D;
    if (a.stage != 0) {
      n += 1;
      for (long long i = lane; i < a.D; i += 64) {
        const double v = a.position[c * a.D + i];
        double mean = a.s.wc_mean[c * a.D + i];
        const double delta = v - mean;
        mean = mean + delta / (double)n;
        a.s.wc_mean[c * a.D + i] = mean;
        fl_delta[i] = delta;
        fl_ud[i] = v - mean;
      }
      __threadfence_block();
      if (a.D <= AEHMC_PC_LDS_MAX_D) {
        for (long long idx = lane; idx < DD; idx += 64)
          a.s.wc_m2[c * DD + idx] = a.s.wc_m2[c * DD + idx] + fl_ud[idx / a.D] * fl_delta[idx % a.D];
      } else {  // row by row: coalesced, no index divisions
        for (long long i = 0; i < a.D; i++) {
          const double ud = fl_ud[i];
          double *m2 = a.s.wc_m2 + c * DD + i * a.D;
          for (long long j = lane; j < a.D; j += 64) m2[j] = m2[j] + ud * fl_delta[j];
        }
      }
    }
    if (a.window_end) {
      const double nn = (double)n;
      const bool in_lds = a.D <= AEHMC_PC_LDS_MAX_D;
      double *A = in_lds ? ad_lds + 2 * a.D : a.s.work + c * DD;
      double *S = in_lds ? A + DD : a.s.sqrt_mass + c * DD;
      for (long long idx = lane; idx < DD; idx += 64) {
        const double cov = a.s.wc_m2[c * DD + idx] / (double)(n - 1);
        double imm = (nn / (nn + 5)) * cov;
        if (idx / a.D == idx % a.D) imm = imm + 1e-3 * (5 / (nn + 5));  // shrinkage * eye
        a.s.imm[c * DD + idx] = imm;
        A[idx] = imm;
        a.s.wc_m2[c * DD + idx] = 0.0;
      }
      for (long long i = lane; i < a.D; i += 64) a.s.wc_mean[c * a.D + i] = 0.0;
      __threadfence_block();
      wave_chol_inv_t(A, S, (int)a.D, lane);  // a non-PD estimate leaves NaNs, as the reference's cholesky would
      if (in_lds)
        for (long long idx = lane; idx < DD; idx += 64) a.s.sqrt_mass[c * DD + idx] = S[idx];
      n = 0;
      adapt_da_restart(da, step_size);
    }
  } else {
  if (a.stage != 0) {  // Welford update with the new position (algorithms.py:187-197)
    n += 1;
    for (long long i = lane; i < a.D; i += 64) {
      double mean = a.s.wc_mean[c * a.D + i], m2 = a.s.wc_m2[c * a.D + i];
      adapt_welford_elem(a.position[c * a.D + i], n, mean, m2);
      a.s.wc_mean[c * a.D + i] = mean;
      a.s.wc_m2[c * a.D + i] = m2;
    }
  }
  if (a.window_end) {  // slow_final: window_adaptation.py:165-182, mass_matrix.py:83-118
    for (long long i = lane; i < a.D; i += 64) {
      double mean = a.s.wc_mean[c * a.D + i], m2 = a.s.wc_m2[c * a.D + i], imm, sqrt_mass;
      adapt_window_end_elem(n, mean, m2, imm, sqrt_mass);
      a.s.imm[c * a.D + i] = imm;
      a.s.sqrt_mass[c * a.D + i] = sqrt_mass;
      a.s.wc_mean[c * a.D + i] = mean;
      a.s.wc_m2[c * a.D + i] = m2;
    }
    n = 0;
    adapt_da_restart(da, step_size);
  }
  }
  if (a.last) step_size = exp(da.x_avg);  // window_adaptation.py:184-190
  if (lane == 0) {
    a.s.da_step[c] = da.step;
    a.s.da_x[c] = da.x;
    a.s.da_x_avg[c] = da.x_avg;
    a.s.da_g_avg[c] = da.g_avg;
    a.s.da_mu[c] = da.mu;
    a.s.wc_n[c] = n;
    a.s.step_size[c] = step_size;
  }
}

// step_size.dual_averaging_adaptation's update alone (step_size.py:97-98, algorithms.py:104-115), one
// thread per chain: the arithmetic of adapt_da_update, hence the bits of the warm-up kernels
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_dual_averaging(long long C, double target, double gamma, double t0,
                                                         double kappa, const double *p_accept, long long *step,
                                                         double *x, double *x_avg, double *g_avg, const double *mu,
                                                         double *step_size_out) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  DualAvg da = {step[c], x[c], x_avg[c], g_avg[c], mu[c]};
  const double eps = adapt_da_update(da, target, p_accept[c], gamma, t0, kappa);
  step[c] = da.step;
  x[c] = da.x;
  x_avg[c] = da.x_avg;
  g_avg[c] = da.g_avg;
  if (step_size_out) step_size_out[c] = eps;
}

// ---- the adaptation building blocks on their own (algorithms.py:120-204, mass_matrix.py:83-118) ----
// algorithms.welford_covariance's update for C independent estimators, one wavefront each: the arithmetic of
// adapt_welford_elem / k_adapt_update's full branch, hence the bits of the warm-up kernels.  full: m2 is [C,D,D] and
// grows by outer(updated_delta, delta) (LDS: delta [D], updated delta [D]).
AEHMC_TU_LOCAL __global__ __launch_bounds__(64) void k_welford_update(long long C, long long D, int full, const double *value,
                                                        double *mean, double *m2, long long *n_arr) {
  extern __shared__ __attribute__((aligned(16))) double wf_lds[];
  const int lane = threadIdx.x;
  const long long c = blockIdx.x;
  const long long n = n_arr[c] + 1;
  if (!full) {
    for (long long i = lane; i < D; i += 64) {
      double mu = mean[c * D + i], s = m2[c * D + i];
      adapt_welford_elem(value[c * D + i], n, mu, s);
      mean[c * D + i] = mu;
      m2[c * D + i] = s;
    }
  } else {
    double *const fl_delta = wf_lds, *const fl_ud = wf_lds + D;
    for (long long i = lane; i < D; i += 64) {
      const double v = value[c * D + i];
      double mu = mean[c * D + i];
      const double delta = v - mu;
      mu = mu + delta / (double)n;
      mean[c * D + i] = mu;
      fl_delta[i] = delta;
      fl_ud[i] = v - mu;
    }
    __threadfence_block();
    for (long long i = 0; i < D; i++) {
      const double ud = fl_ud[i];
      double *row = m2 + c * D * D + i * D;
      for (long long j = lane; j < D; j += 64) row[j] = row[j] + ud * fl_delta[j];
    }
  }
  if (lane == 0) n_arr[c] = n;
}
// welford_covariance's final (algorithms.py:199-202: m2 / (n - 1)) and, with `shrink`, covariance_adaptation's final
// (mass_matrix.py:83-118: Stan's shrinkage towards 1e-3 -- on every element of a diagonal estimate, on the diagonal
// of a full one), one thread per element; the expressions of adapt_window_end_elem
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_covariance_final(long long C, long long per, long long D, int full, int shrink,
                                                           const double *m2, const long long *n_arr, double *out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C * per) return;
  const long long c = e / per, idx = e % per;
  const long long n = n_arr[c];
  const double nn = (double)n;
  const double cov = m2[e] / (double)(n - 1);
  double r = cov;
  if (shrink) {
    r = (nn / (nn + 5)) * cov;
    if (!full || idx / D == idx % D) r = r + 1e-3 * (5 / (nn + 5));
  }
  out[e] = r;
}

AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_fill_i64(long long *x, long long n, long long v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = v;
}
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_add_i64(long long *acc, const long long *x, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long long)gridDim.x * blockDim.x) acc[i] += x[i];
}
// ---- gaussian_metric set-up (metrics.py:44-59): sqrt(1/imm), or L^-T with imm = L L^T ----
AEHMC_TU_LOCAL __global__ void k_sqrt_recip(const double *x, double *y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sqrt(1.0 / x[i]);
}
constexpr int FACT_NB = 64;
// One workgroup: (optionally) Cholesky-factor the n x n (n <= 64) diagonal block in place
// (lower), then invert the lower-triangular factor; writes inv and inv^T, zero padded to
// [64][64].  info: first non-positive pivot (1-based, global index) if any.
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_potrf_trtri(double *A, long long ld, int n, int do_factor,
                                                     double *inv, double *invT, int *info, int pivot_base) {
  __shared__ double L[FACT_NB][FACT_NB + 1], X[FACT_NB][FACT_NB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < FACT_NB * FACT_NB; e += 256) {
    const int i = e / FACT_NB, j = e % FACT_NB;
    L[i][j] = (i < n && j <= i) ? A[(long long)i * ld + j] : 0.0;
    X[i][j] = 0.0;
  }
  __syncthreads();
  if (do_factor) {
    for (int j = 0; j < n; j++) {
      if (tid == 0) {
        const double d = L[j][j];
        if (!(d > 0.0) && *info == 0) *info = pivot_base + j + 1;
        L[j][j] = sqrt(d);
      }
      __syncthreads();
      const double djj = L[j][j];
      for (int i = j + 1 + tid; i < n; i += 256) L[i][j] = L[i][j] / djj;
      __syncthreads();
      for (int e = tid; e < FACT_NB * FACT_NB; e += 256) {
        const int i = e / FACT_NB, k = e % FACT_NB;
        if (k > j && k <= i && i < n) L[i][k] = L[i][k] - L[i][j] * L[k][j];
      }
      __syncthreads();
    }
    for (int e = tid; e < FACT_NB * FACT_NB; e += 256) {
      const int i = e / FACT_NB, j = e % FACT_NB;
      if (i < n && j <= i) A[(long long)i * ld + j] = L[i][j];
    }
  }
  if (tid < n) {  // column tid of L^-1 by forward substitution
    const int t = tid;
    for (int i = t; i < n; i++) {
      double s = (i == t) ? 1.0 : 0.0;
      for (int j = t; j < i; j++) s -= L[i][j] * X[j][t];
      X[i][t] = s / L[i][i];
    }
  }
  __syncthreads();
  for (int e = tid; e < FACT_NB * FACT_NB; e += 256) {
    const int i = e / FACT_NB, j = e % FACT_NB;
    inv[e] = X[i][j];
    invT[e] = X[j][i];
  }
}
AEHMC_TU_LOCAL __global__ void k_copy_block(const double *src, long long lds_, double *dst, long long ldd, int rows, int cols) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += gridDim.x * blockDim.x) {
    const int i = e / cols, j = e % cols;
    dst[(long long)i * ldd + j] = src[(long long)i * lds_ + j];
  }
}
AEHMC_TU_LOCAL __global__ void k_transpose(const double *src, double *dst, long long n) {  // dst = src^T, n x n
  __shared__ double tile[32][33];
  const long long bx = (long long)blockIdx.x * 32, by = (long long)blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const long long i = by + r, j = bx + threadIdx.x;
    tile[r][threadIdx.x] = (i < n && j < n) ? src[i * n + j] : 0.0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const long long i = bx + r, j = by + threadIdx.x;
    if (i < n && j < n) dst[i * n + j] = tile[threadIdx.x][r];
  }
}
AEHMC_TU_LOCAL __global__ void k_log(const double *x, double *y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = log(x[i]);
}

}  // namespace aehmc
