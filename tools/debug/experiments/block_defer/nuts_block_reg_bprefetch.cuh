// Block-resident dense NUTS / HMC with the chains' moving state in REGISTERS (gfx950): 64 < D <= 256.
//
// nuts_block.cuh runs the lock-step engine's own stage / bookkeeping functions on L2-resident work rows between the
// in-workgroup MFMA products; counted per leapfrog of a workgroup at D = 200 (tools/debug/block_phases.py), those
// stages -- a dozen dependent L2 round trips, 200+ bytes of scratch per lane at the 128 registers a 1024-thread
// workgroup leaves a lane -- cost as much as the two products.  Here the wavefront that owns a chain keeps q, p,
// dU/dq, the velocity v = imm p, w = imm dU/dq and the sub-trajectory momentum sum in VGPRs, element lane + 64 r in
// slot r (D <= 64 R), exactly as k_nuts_resident does for diagonal metrics; the products read their operand rows
// from one LDS buffer and write the result rows to another, which the owners read back -- the chain state touches
// global memory only where the tree needs it (U-turn checkpoints, trajectory ends at expansion boundaries, the
// proposal on accept).  Per leapfrog: first stages in registers -> operand row to LDS -> P r -> imm g' (LDS to LDS)
// -> rows back -> last stage + bookkeeping in registers; three workgroup barriers.
// Same arithmetic in the same order as engine.cuh's leap_linear / nuts_book / nuts_finalize_expansion / hmc_end_chain
// (each lane adds its elements in ascending order, sums by wave_sum), same MFMA k-chains: BITWISE the lock-step
// path's results (tests/test_gpu_block_dense.py).
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235, proposals.py:19-174,
// hmc.py:77-204, integrators.py:54-73, metrics.py:44-104.
#pragma once
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime itself) */
#include <hip/hip_runtime.h>
#endif

#include "nuts_block.cuh"

namespace aehmc {

constexpr int BLK_REG_MAX_D = 256;
constexpr int BLK_PARK = 20;  // doubles per wavefront: the chain's four generators (see k_nuts_block_reg) + the rolling kernel's between-transition scalars
#ifndef __HIPCC_RTC__  // (host side)
inline bool block_reg_supported(long long D) { return D >= BLK_MIN_D && D <= BLK_REG_MAX_D; }
// two row buffers [16][S] (operand / result, swapping roles from product to product) + one staging tile per wavefront
// + the target's mean [S] + the parking areas
inline size_t blk_reg_lds_bytes(long long D) {
  return ((size_t)(2 * BLK_CHAINS + 1) * blk_lds_stride(D) + (size_t)BLK_CHAINS * (BLK_TB + BLK_PARK)) * sizeof(double);
}
#endif  // __HIPCC_RTC__

// the chain's four generators have their home in LDS (d + 4 k): a draw loads ONE of them, advances it and puts it
// back -- 8 registers for the duration of the draw instead of 32 for the duration of the launch
__device__ __forceinline__ Pcg64 blk_gen_load(const double *d, int k) {
  const unsigned long long *u = reinterpret_cast<const unsigned long long *>(d) + 4 * k;
  Pcg64 g;
  g.state = mk128(u[0], u[1]);
  g.inc = mk128(u[2], u[3]);
  return g;
}
__device__ __forceinline__ void blk_gen_store(double *d, int k, const Pcg64 &g, int lane) {
  unsigned long long *u = reinterpret_cast<unsigned long long *>(d) + 4 * k;
  if (lane == 0) {  // (every lane holds the same state)
    u[0] = g.state.hi; u[1] = g.state.lo;
    u[2] = g.inc.hi; u[3] = g.inc.lo;
  }
}
__device__ __forceinline__ int blk_bernoulli(double *d, int k, double p, int lane) {
  Pcg64 g = blk_gen_load(d, k);
  const int r = rng_bernoulli(g, p);
  blk_gen_store(d, k, g, lane);
  return r;
}

// A workgroup barrier that orders LDS traffic only.  The wavefronts of this kernel talk to each other through LDS alone
// (operand / result rows, status words); what a chain keeps in global memory is never read by another chain.
// __syncthreads() would also wait for every outstanding GLOBAL access of the wavefront (s_waitcnt vmcnt(0)): the loads
// that are requested a phase ahead on purpose would be waited for at the very next barrier, by the whole workgroup.
__device__ __forceinline__ void blk_barrier_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// a value the optimizer has to take as it comes (see ATL in nuts_block_roll.cuh)
__device__ __forceinline__ int blk_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// dst[16][S] = src[16][S] * Bp^T, both in LDS, for the rows of `rowmask`; the caller places the barriers
__device__ __forceinline__ void blk_gemm_lds(const double *src, double *dst, int S, const double *Bp, long long D,
                                             int wave, int lane, double *tb, unsigned rowmask = 0xffffu) {
  const int NT = (int)((D + 15) / 16);
  for (int nt = wave; nt < NT; nt += BLK_CHAINS) blk_wave_tile(src, S, Bp, NT * 16, D, nt * 16, dst, S, rowmask, lane, tb);
}
// the same with the head of the wavefront's column block of B requested by blk_gemm_lds_prefetch (before the barrier
// in front of the product): D <= BLK_REG_MAX_D, so a wavefront has at most one block
__device__ __forceinline__ void blk_gemm_lds_prefetch(const double *Bp, long long D, int wave, int lane, BlkPre &pre) {
  const int NT = (int)((D + 15) / 16);
  if (wave < NT) blk_tile_prefetch(Bp, NT * 16, wave * 16, lane, pre);
}
__device__ __forceinline__ void blk_gemm_lds_pre(const double *src, double *dst, int S, const double *Bp, long long D,
                                                 int wave, int lane, double *tb, const BlkPre &pre, unsigned rowmask = 0xffffu) {
  const int NT = (int)((D + 15) / 16);
  if (wave < NT) blk_wave_tile<true>(src, S, Bp, NT * 16, D, wave * 16, dst, S, rowmask, lane, tb, &pre);
}

template <int R, bool TDENSE>
__global__ __launch_bounds__(BLK_THREADS) void k_nuts_block_reg(EngineArgs a, NutsSampleArgs m) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS, c = c0 + wave;
  const bool valid = c < a.C;
  const long long D = a.D;
  const int S = (int)blk_lds_stride(D);
  double *const xbuf = blk_lds, *const ybuf = blk_lds + BLK_CHAINS * S;
  double *const tb = blk_lds + 2 * BLK_CHAINS * S + wave * BLK_TB;
  double *const mus = blk_lds + 2 * BLK_CHAINS * S + BLK_CHAINS * BLK_TB;  // the dense target's mean
  // Register budget: 16 wavefronts per workgroup leave a lane 128 registers.  The chain's q, p, v and momentum sum
  // (8 R registers), its tree state (wave-uniform, but the product of fp64 VALU arithmetic: ~26 VGPRs) and the
  // product's ~45 fit; dU/dq and w do not have to be in registers (they stay in the LDS rows the products write), and
  // the four generators (32 more: 64-bit integer VALU arithmetic) have their home in LDS, a draw holding one of them
  // for its own duration.  A first version with q, p, g, v, w and the generators in registers spilled 270-450 bytes
  // per lane: the spill code alone moved 2.9 TB/s through HBM and the kernel waited on it (profiles/r4/INDEX.md);
  // now the per-leapfrog loop has no scratch access.
  double *const park = mus + S + wave * BLK_PARK;
  double *const xrow = xbuf + wave * S, *const yrow = ybuf + wave * S;
  const size_t row = (size_t)(valid ? c : 0) * D;
  const bool elem = target_is_elem(a.tkind);
  for (int k = lane; k < S; k += 64) {  // pads (and the rows of chains past C) stay zero for the whole launch
    xrow[k] = 0.0;
    yrow[k] = 0.0;
  }
  if (TDENSE)
    for (int k = threadIdx.x; k < S; k += BLK_THREADS) mus[k] = k < D ? a.mu[k] : 0.0;
#define EI(r) (lane + 64 * (r))
#define AT(ptr, r) ((ptr) + row)[EI(r)]
  bool ok[R];
  // q, p, v and the sub-trajectory momentum sum in registers; dU/dq and w = imm dU/dq stay where the products leave
  // them, in the chain's rows of the two LDS buffers (dense target: P r lands in ybuf, imm g' in xbuf; coordinate-wise
  // target: g' is written to xbuf as the operand, imm g' lands in ybuf) -- 16 registers less at R = 4
  double q[R], p[R], v[R], pb[R];
  double *const grow = TDENSE ? yrow : xrow, *const wrow = TDENSE ? xrow : yrow;
#pragma unroll
  for (int r = 0; r < R; r++) {
    ok[r] = valid && EI(r) < D;
    q[r] = p[r] = v[r] = pb[r] = 0.0;
  }
  ChainCtl ct = {};
  ct.done = 1;
  double U_state = 0.0;
  long long nleap_sum = 0;
  double eps = 0.0;
  if (valid) {
    const ChainRng rng0 = rng_load(a, c);
#pragma unroll
    for (int k = 0; k < 4; k++) blk_gen_store(park, k, rng0.g[k], lane);
    U_state = a.U[c];
    eps = a.eps_c ? a.eps_c[c] : a.eps;
  }
  BlkTimer tm;
  __shared__ int blk_alive[BLK_CHAINS];  // the round's vote (the barriers order LDS traffic only: blk_barrier_lds)

#define BT_LC lane
#define BTA(ptr, r) ((ptr) + row)[lc + 64 * (r)]
#define BT_SLOT_Q(s) pick2(a.slot_q, s)
#define BT_SLOT_P(s) pick2(a.slot_p, s)
#define BT_SLOT_G(s) pick2(a.slot_g, s)
#define BT_END_Q(e) pick2(a.end_q, e)
#define BT_END_P(e) pick2(a.end_p, e)
#define BT_END_G(e) pick2(a.end_g, e)
#define BT_END_V(e) pick2(a.end_v, e)
#define BT_END_W(e) pick2(a.end_w, e)
#define BT_PSUM a.psum
#define BT_TAKE_W(slot, r) (void)0
#define BT_OUT_Q a.q
#define BT_OUT_G a.g
#define BT_OUT_U a.U
#define BT_OUT_MOM a.out.momentum
#define BT_OUT_ACC a.out.acceptance_probability
#define BT_OUT_NDOUBL a.out.num_doublings
#define BT_OUT_TURN a.out.is_turning
#define BT_OUT_DIV a.out.is_diverging
#define BT_OUT_NLEAP a.out.n_leapfrog
#include "nuts_block_tree.inc"
#undef BT_LC
#undef BTA
#undef BT_SLOT_Q
#undef BT_SLOT_P
#undef BT_SLOT_G
#undef BT_END_Q
#undef BT_END_P
#undef BT_END_G
#undef BT_END_V
#undef BT_END_W
#undef BT_PSUM
#undef BT_TAKE_W
#undef BT_OUT_Q
#undef BT_OUT_G
#undef BT_OUT_U
#undef BT_OUT_MOM
#undef BT_OUT_ACC
#undef BT_OUT_NDOUBL
#undef BT_OUT_TURN
#undef BT_OUT_DIV
#undef BT_OUT_NLEAP

  blk_barrier_lds();
  for (long long t_idx = 0; t_idx < m.T; t_idx++) {
    // ---- momentum: z (site #1) -> p = L^-T z, v = imm p; w = imm dU/dq (metrics.py:65-68, nuts.py:113-125) ----
    if (valid) {
      Pcg64 g0 = blk_gen_load(park, 0);
      wave_normals(g0, D, [=](long long i, double z) { xrow[i] = z; });
      blk_gen_store(park, 0, g0, lane);
      __threadfence_block();
    }
    tm.tick(7);
    blk_barrier_lds();
    blk_gemm_lds(xbuf, ybuf, S, a.sqrt_mass, D, wave, lane, tb);
    blk_barrier_lds();
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = ok[r] ? yrow[EI(r)] : 0.0;
    blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb);
    blk_barrier_lds();
#pragma unroll
    for (int r = 0; r < R; r++) {
      v[r] = ok[r] ? xrow[EI(r)] : 0.0;
      q[r] = ok[r] ? AT(a.q, r) : 0.0;
      pb[r] = 0.0;
      if (ok[r]) grow[EI(r)] = AT(a.g, r);  // the operand row of w = imm dU/dq, and dU/dq's home from here on
    }
    blk_barrier_lds();
    if (TDENSE) blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb);
    else blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb);
    blk_barrier_lds();
    tm.tick(2);
    if (valid) {  // nuts_init_chain<true>
      double kd = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double gr = grow[EI(r)], wr = wrow[EI(r)];
          kd += v[r] * p[r];
#pragma unroll
          for (int e = 0; e < 2; e++) {
            AT(a.end_q[e], r) = q[r];
            AT(a.end_p[e], r) = p[r];
            AT(a.end_g[e], r) = gr;
            AT(a.end_v[e], r) = v[r];
            AT(a.end_w[e], r) = wr;
          }
          AT(a.slot_q[0], r) = q[r];
          AT(a.slot_p[0], r) = p[r];
          AT(a.slot_g[0], r) = gr;
          AT(a.psum, r) = p[r];
        }
      }
      kd = wave_sum(kd);
      const double U = U_state;
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
      ct.dir = blk_bernoulli(park, 1, 0.5, lane);  // trajectory.py:516
      ct.step = 0;
    }
    tm.tick(7);
    // ---- one leapfrog of every live chain per trip: first stages | P r | imm g' | last stage + bookkeeping ----
    // (the head of each product's matrix block is requested before the barrier in front of the product: BlkPre)
    for (;;) {
      if (valid && !ct.done) stage12();
      tm.tick(5);
      const bool alive = valid && !ct.done;
      if (lane == 0) blk_alive[wave] = alive ? 1 : 0;
      BlkPre pre;
      blk_gemm_lds_prefetch(TDENSE ? m.prec : a.imm, D, wave, lane, pre);
      blk_barrier_lds();
      int live = 0;
#pragma unroll
      for (int k = 0; k < BLK_CHAINS; k++) live |= blk_alive[k];
      live = __builtin_amdgcn_readfirstlane(live);
      tm.tick(6);  // vote (waits for the slowest chain's bookkeeping)
      if (!live) break;
      if (TDENSE) {
        blk_gemm_lds_pre(xbuf, ybuf, S, m.prec, D, wave, lane, tb, pre);  // dU/dq' = P r
        blk_gemm_lds_prefetch(a.imm, D, wave, lane, pre);
        tm.tick(2);
        blk_barrier_lds();
        tm.tick(3);
        blk_gemm_lds_pre(ybuf, xbuf, S, a.imm, D, wave, lane, tb, pre);   // w' = imm dU/dq'
        tm.tick(2);
        blk_barrier_lds();
        tm.tick(3);
      } else {
        blk_gemm_lds_pre(xbuf, ybuf, S, a.imm, D, wave, lane, tb, pre);   // w' = imm dU/dq'
        tm.tick(2);
        blk_barrier_lds();
        tm.tick(3);
      }
      if (alive) {
        book();
        tm.tick(4);
      }
    }
    // ---- per-transition records (the outputs themselves were written when the transition ended) ----
    if (valid) {
      U_state = pick2(ct.U_slot, ct.prop_slot);
      nleap_sum += ct.nleap;
      if (m.samples) {
        double *dst = m.samples + ((size_t)t_idx * a.C + c) * D;
#pragma unroll
        for (int r = 0; r < R; r++)
          if (ok[r]) dst[EI(r)] = AT(a.q, r);
      }
      if (lane == 0) {
        if (m.acc_hist) m.acc_hist[(size_t)t_idx * a.C + c] = ct.acc_prob;
        if (m.div_hist) m.div_hist[(size_t)t_idx * a.C + c] = ct.out_div;
      }
    }
  }
  if (valid) {
    ChainRng rng1;
#pragma unroll
    for (int k = 0; k < 4; k++) rng1.g[k] = blk_gen_load(park, k);
    rng_store(a, c, lane, rng1, 0, 3);
    if (lane == 0 && m.nleap_total) m.nleap_total[c] = nleap_sum;
#ifdef AEHMC_WIDE_TIMING
    if (lane == 0)
      for (int k = 0; k < 16; k++) a.linreg_part[c * 16 + k] = (double)tm.acc[k];
#endif
  }
  (void)elem;
#undef EI
#undef AT
}

// HMC: nt transitions x L leapfrogs, the chains' q, p, v in registers, dU/dq and w = imm dU/dq in the LDS rows the
// products write (as above) -- hmc_run's lock-step loop: hmc_init_chain, leap_linear<12> / <3>, hmc_end_chain.  The
// state a rejection falls back to is the caller's q / dU/dq, rewritten at every accepted transition.
template <int R, bool TDENSE>
__global__ __launch_bounds__(BLK_THREADS) void k_hmc_block_reg(EngineArgs a, const double *prec, long long L,
                                                                long long nt, double *samples, double *acc_hist,
                                                                int *div_hist) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS, c = c0 + wave;
  const bool valid = c < a.C;
  const long long D = a.D;
  const int S = (int)blk_lds_stride(D);
  double *const xbuf = blk_lds, *const ybuf = blk_lds + BLK_CHAINS * S;
  double *const tb = blk_lds + 2 * BLK_CHAINS * S + wave * BLK_TB;
  double *const mus = blk_lds + 2 * BLK_CHAINS * S + BLK_CHAINS * BLK_TB;
  double *const xrow = xbuf + wave * S, *const yrow = ybuf + wave * S;
  double *const grow = TDENSE ? yrow : xrow, *const wrow = TDENSE ? xrow : yrow;
  const size_t row = (size_t)(valid ? c : 0) * D;
  for (int k = lane; k < S; k += 64) {
    xrow[k] = 0.0;
    yrow[k] = 0.0;
  }
  if (TDENSE)
    for (int k = threadIdx.x; k < S; k += BLK_THREADS) mus[k] = k < D ? a.mu[k] : 0.0;
#define EI(r) (lane + 64 * (r))
#define AT(ptr, r) ((ptr) + row)[EI(r)]
  bool ok[R];
  double q[R], p[R], v[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    ok[r] = valid && EI(r) < D;
    q[r] = ok[r] ? AT(a.q, r) : 0.0;
    p[r] = v[r] = 0.0;
  }
  Pcg64 g1 = {}, g2 = {};
  double U = 0.0, eps = 0.0, pa = 0.0;
  int is_div = 0, acc = 0;
  if (valid) {
    g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4);
    g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
    U = a.U[c];
    eps = a.eps_c ? a.eps_c[c] : a.eps;
  }
  const double b = 0.5 * (1.0 * eps), aa = 1 * (1.0 * eps);  // direction +1 (launch_leapfrog: ct.dir = 1)
  blk_barrier_lds();
  for (long long tt = 0; tt < nt; tt++) {
    const bool last_t = tt == nt - 1;
    if (valid) {
      wave_normals(g1, D, [=](long long i, double z) { xrow[i] = z; });
      __threadfence_block();
    }
    blk_barrier_lds();
    blk_gemm_lds(xbuf, ybuf, S, a.sqrt_mass, D, wave, lane, tb);  // p = L^-T z
    blk_barrier_lds();
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = ok[r] ? yrow[EI(r)] : 0.0;
    blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb);        // v = imm p
    blk_barrier_lds();
#pragma unroll
    for (int r = 0; r < R; r++) {
      v[r] = ok[r] ? xrow[EI(r)] : 0.0;
      if (ok[r]) {
        grow[EI(r)] = AT(a.g, r);  // the operand row of w = imm dU/dq, and dU/dq's home during the trajectory
        // only the last transition's momentum is observable: the initial one is kept on rejection
        if (last_t && a.out.momentum) AT(a.out.momentum, r) = p[r];
      }
    }
    blk_barrier_lds();
    if (TDENSE) blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb);  // w = imm dU/dq
    else blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb);
    blk_barrier_lds();
    double kd = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++)  // hmc_init_chain<true>
      if (ok[r]) kd += v[r] * p[r];
    kd = wave_sum(kd);
    const double H0 = U + 0.5 * kd;  // hmc.py:187
    double U_cur = U;
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95: leap_linear<12> | P r | imm g' | leap_linear<3>
      double usum = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double pp = p[r] - b * grow[EI(r)];
          const double vv = v[r] - b * wrow[EI(r)];
          p[r] = pp;
          v[r] = vv;
          const double qq = q[r] + aa * vv;
          q[r] = qq;
          if (!TDENSE) {
            double u, gnew;
            target_elem(a, EI(r), qq, u, gnew);
            usum += u;
            xrow[EI(r)] = gnew;
          } else {
            xrow[EI(r)] = qq - mus[EI(r)];
          }
        }
      }
      if (!TDENSE && valid) U_cur = target_finish(a, wave_sum(usum));
      blk_barrier_lds();
      if (TDENSE) {
        blk_gemm_lds(xbuf, ybuf, S, prec, D, wave, lane, tb);
        blk_barrier_lds();
        blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb);
        blk_barrier_lds();
      } else {
        blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb);
        blk_barrier_lds();
      }
      usum = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double gr = grow[EI(r)], wr = wrow[EI(r)];
          if (TDENSE) usum += (q[r] - mus[EI(r)]) * gr;
          p[r] = p[r] - b * gr;
          v[r] = v[r] - b * wr;
        }
      }
      if (TDENSE && valid) U_cur = target_finish(a, wave_sum(usum));
    }
    if (valid) {  // hmc_end_chain<true>
      kd = 0.0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (ok[r]) {
          const double pf = -1.0 * p[r];  // hmc.py:185 momentum flip
          const double vf = -1.0 * v[r];
          kd += vf * pf;
        }
      }
      kd = wave_sum(kd);
      const double new_energy = U_cur + 0.5 * kd;
      double delta = H0 - new_energy;
      if (isnan(delta)) delta = -INFINITY;
      is_div = fabs(delta) > a.thr;
      pa = exp(delta);
      if (pa > 1.0) pa = 1.0;
      if (pa < 0.0) pa = 0.0;
      acc = rng_bernoulli(g2, pa);  // hmc.py:193-194
      if (acc) {  // the caller's arrays follow every accepted transition (they are the state a rejection falls back to)
        U = U_cur;
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (ok[r]) {
            AT(a.q, r) = q[r];
            AT(a.g, r) = grow[EI(r)];
            if (last_t && a.out.momentum) AT(a.out.momentum, r) = -1.0 * p[r];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) q[r] = ok[r] ? AT(a.q, r) : 0.0;
      }
      if (samples) {
        double *dst = samples + ((size_t)tt * a.C + c) * D;
#pragma unroll
        for (int r = 0; r < R; r++)
          if (ok[r]) dst[EI(r)] = q[r];
      }
      if (lane == 0) {
        if (acc_hist) acc_hist[(size_t)tt * a.C + c] = pa;
        if (div_hist) div_hist[(size_t)tt * a.C + c] = is_div;
      }
    }
  }
  if (valid && lane == 0) {
    pcg_store(a.rng + (size_t)c * a.nsites * 4, g1);
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, g2);
    a.U[c] = U;
    a.out.acceptance_probability[c] = pa;
    a.out.is_diverging[c] = is_div;
    if (a.out.n_leapfrog) a.out.n_leapfrog[c] = L;
    if (a.out.is_turning) a.out.is_turning[c] = acc;  // HMC: reused as the accept flag
  }
#undef EI
#undef AT
}

#ifndef __HIPCC_RTC__  // (host side)
template <int R>
inline hipError_t launch_nuts_block_reg_r(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) {
  const size_t dyn = blk_reg_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BLK_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_block_reg<R, TDV>),            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_nuts_block_reg<R, TDV>), grid, block, dyn, st, a, m);                              \
  } while (0)
  if (a.tkind == AEHMC_T_DENSE_MVN) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}
inline hipError_t launch_nuts_block_reg(EngineArgs a, NutsSampleArgs m, double *bp, hipStream_t st) {
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, m.prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass; m.prec = mats.prec;
  return a.D <= 128 ? launch_nuts_block_reg_r<2>(a, m, st) : launch_nuts_block_reg_r<4>(a, m, st);
}

template <int R>
inline hipError_t launch_hmc_block_reg_r(const EngineArgs &a, const double *prec, long long L, long long nt,
                                         double *samples, double *acc_hist, int *div_hist, hipStream_t st) {
  const size_t dyn = blk_reg_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BLK_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hmc_block_reg<R, TDV>),             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_hmc_block_reg<R, TDV>), grid, block, dyn, st, a, prec, L, nt, samples, acc_hist,   \
                       div_hist);                                                                            \
  } while (0)
  if (a.tkind == AEHMC_T_DENSE_MVN) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}
inline hipError_t launch_hmc_block_reg(EngineArgs a, const double *prec, long long L, long long nt, double *samples,
                                       double *acc_hist, int *div_hist, double *bp, hipStream_t st) {
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass;
  return a.D <= 128 ? launch_hmc_block_reg_r<2>(a, mats.prec, L, nt, samples, acc_hist, div_hist, st)
                    : launch_hmc_block_reg_r<4>(a, mats.prec, L, nt, samples, acc_hist, div_hist, st);
}
#endif  // __HIPCC_RTC__

}  // namespace aehmc
