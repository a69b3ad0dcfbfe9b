// Block-resident dense NUTS, the chains of a workgroup FLOWING through their transitions (gfx950): 64 < D <= 256.
//
// Round 4 had two kernels here: k_nuts_block_reg ran the 16 chains of a workgroup transition by transition (every
// transition as long as the deepest of 16 trees: 27 rounds at D = 200 where the mean tree has 17 leapfrogs), and
// k_nuts_block_roll let finished chains begin again once a few of them waited -- but a beginning chain needed three
// products of its own, which the round had to carry for everybody, and a register-hungry normal draw inside the round
// loop; the shorter schedule arrived as 2-5 % of wall time, and as a loss below D = 192.  Both gave a chain a whole
// wavefront, 16 wavefronts per workgroup: 128 registers per lane (108-216 bytes of scratch), and -- what the round-5
// census of the bookkeeping phase showed (tools/debug/block_phases.py CENSUS=1) -- every wavefront evaluating ITS chain's
// scalar code (energies, exp / log1p of the proposal weights, the PCG draws, the tree's state machine) on 64 redundant
// lanes at 4 issue cycles per instruction, four wavefronts per SIMD taking turns on one vector ALU: the bookkeeping of a
// round took 11-18 thousand cycles at D = 100 (28-57 thousand at D = 200 with its spills) against 4.4 (34) thousand for
// the two products, and the workgroup waited for the slowest chain in every round.  This kernel (round 5) replaces both:
//   * A chain is a TEAM OF 16 LANES -- one DPP row -- four chains per wavefront, four wavefronts (256 threads) per
//     workgroup of 16 chains.  The per-chain scalar code is ordinary SIMT code, evaluated once per wavefront for four
//     chains; a lane has the whole register file of a one-wavefront-per-SIMD kernel (the chain's q, p, v, momentum sum
//     and three prefetch vectors of D / 16 elements each stay in registers up to D = 256, nothing spills); sums over D
//     are four DPP row reductions -- lane l of the team keeps the four partial sums of the 64-lane layout's lanes
//     l, l + 16, l + 32, l + 48 (element e belongs to lane e mod 64), each reduced over its row by the same four
//     butterfly stages and combined ((r0 + r16) + r32) + r48: the bits of engine.cuh's wave_sum.
//   * Everything a transition needs BEFORE its tree depends only on the momentum stream (site #1 serves nothing else)
//     and on the launch's initial state: the host side (engine.hip, block_flow_run) draws the normals of all
//     transitions of the launch with k_draw_momentum and forms P = Z L^-1 (p = L^-T z, metrics.py:66-67), V = P imm and
//     w0 = dU/dq imm as chain-batched GEMMs over (transitions x chains) rows -- full MFMA tiles instead of 16-row ones,
//     the same k-chains, the same bits.  A chain whose tree has ended requests its next state, sits out ONE round
//     while the loads arrive, and goes on: no thresholds, no extra products, no draw in this kernel.
//   * Memory reads of the bookkeeping are requested a phase ahead: the first level of the iterative U-turn check is the
//     pair the previous (even) step stored, kept in registers (`pf`) -- except after the stale indices of a
//     sub-trajectory's step 0 (termination.py:109-113), where it is fetched during the products; the other end's
//     momentum / velocity and the momentum sum are fetched during the products of a sub-trajectory's last step; the
//     second U-turn level and the state a change of direction continues from are requested before the scalar work that
//     decides whether they are needed.  The workgroup barriers order LDS traffic only (blk_barrier_lds).
//   * The products: a wavefront computes its column blocks (every fourth) TOGETHER -- the A fragments of a K-tile are
//     read once for all of them and their MFMA chains interleave (one wavefront per SIMD: nobody else fills the pipe).
// Arithmetic, order of operations and RNG consumption per chain are those of engine.cuh's leap_linear / nuts_book /
// nuts_finalize_expansion: BITWISE the lock-step path's results whatever the schedule (tests/test_gpu_block_dense.py).
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235, proposals.py:19-174,
// integrators.py:54-73, metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "nuts_block_reg.cuh"

namespace aehmc {

struct BlkFlowArgs {
  const double *p_all, *v_all;  // [nt][C][D] momenta p = L^-T z and velocities v = imm p of the launch's transitions
  const double *w0;             // [C][D] imm dU/dq of the state the launch starts from
  long long t0, nt;             // this launch runs transitions t0 .. t0 + nt - 1 of the call (record indices)
};

constexpr int BQ_WAVES = 4;                  // wavefronts per workgroup
constexpr int BQ_THREADS = 64 * BQ_WAVES;    // 16 chains x 16 lanes
constexpr int BQ_TILES = 4;                  // column blocks a wavefront computes together (16 blocks at D = 256)
inline size_t blk_flow_lds_bytes(long long D) {  // two row buffers, the staging tiles, the mean, the generators
  return ((size_t)(2 * BLK_CHAINS + 1) * blk_lds_stride(D) + (size_t)BQ_WAVES * BQ_TILES * BLK_TB + (size_t)BLK_CHAINS * BLK_PARK) * sizeof(double);
}

// sum over one DPP row (16 lanes), every lane of the row ends with the same bits: the row stages of wave_sum
__device__ __forceinline__ double row_allsum(double x) {
  x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);  // row_half_mirror
  x = dpp_add<0x140>(x);  // row_mirror
  return x;
}
// A 16-lane team's sum over D.  pt[k]: this lane's sum of its elements tl + 16 (k + 4 m), m ascending -- what lane
// tl + 16 k of a 64-lane wavefront accumulates in engine.cuh; rows combined in wave_sum's order.
__device__ __forceinline__ double team_sum(const double (&pt)[4]) {
  const double r0 = row_allsum(pt[0]), r1 = row_allsum(pt[1]), r2 = row_allsum(pt[2]), r3 = row_allsum(pt[3]);
  return ((r0 + r1) + r2) + r3;
}
// lane k of this lane's row
__device__ __forceinline__ double row_bcast(double x, int k, int lane) {
  const int idx = ((lane & 48) | k) << 2;
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(idx, __double2hiint(x)),
                          __builtin_amdgcn_ds_bpermute(idx, __double2loint(x)));
}
// nuts_step_scalars / nuts_expansion_scalars (engine.cuh) for a team: the same per-lane instruction sequences in lanes
// 0..2 / 0..3 of the ROW, the results handed to the row
__device__ __forceinline__ StepScalars team_step_scalars(double sub_w, double np_w, double sub_slpa, double np_slpa,
                                                         int tl, int lane) {
  const double x = tl == 1 ? sub_w : sub_slpa, y = tl == 1 ? np_w : np_slpa;  // lanes 1, 2: logaddexp(x, y)
  const double tmp = x - y;
  const double earg = tl == 0 ? -(np_w - sub_w) : (tmp > 0 ? -tmp : tmp);
  const double e = exp(earg);
  const double l = log1p(e);
  double la = (x == y) ? x + 0.693147180559945309417232121458176568 : (tmp > 0 ? x + l : (tmp <= 0 ? y + l : tmp));
  double pa = 1.0 / (1.0 + e);
  if (isnan(pa)) pa = 0.0;
  const double r = tl == 0 ? pa : la;
  StepScalars o;
  o.pa = row_bcast(r, 0, lane);
  o.sub_w = row_bcast(r, 1, lane);
  o.sub_slpa = row_bcast(r, 2, lane);
  return o;
}
__device__ __forceinline__ ExpansionScalars team_expansion_scalars(double sub_w, double prop_w, double sub_slpa,
                                                                   double prop_slpa, bool swap, int tl, int lane) {
  const double x = tl == 2 ? prop_w : (swap ? sub_slpa : prop_slpa);   // lanes 2, 3: logaddexp(x, y)
  const double y = tl == 2 ? sub_w : (swap ? prop_slpa : sub_slpa);
  const double tmp = x - y;
  const double earg = tl == 0 ? sub_slpa : tl == 1 ? sub_w - prop_w : (tmp > 0 ? -tmp : tmp);
  const double e = exp(earg);
  const double l = log1p(e);
  const double la = (x == y) ? x + 0.693147180559945309417232121458176568 : (tmp > 0 ? x + l : (tmp <= 0 ? y + l : tmp));
  const double r = tl < 2 ? e : la;
  ExpansionScalars o;
  o.e_slpa = row_bcast(r, 0, lane);
  o.e_ratio = row_bcast(r, 1, lane);
  o.la_w = row_bcast(r, 2, lane);
  o.la_slpa = row_bcast(r, 3, lane);
  return o;
}
// Generator.binomial(1, p) from the chain's generator k (home in LDS at d + 4 k; every lane of the team computes the
// same draw, its lane 0 puts the state back)
__device__ __forceinline__ int team_bernoulli(double *d, int k, double p, int tl) {
  Pcg64 g = blk_gen_load(d, k);
  const int r = rng_bernoulli(g, p);
  unsigned long long *u = reinterpret_cast<unsigned long long *>(d) + 4 * k;
  if (tl == 0) {
    u[0] = g.state.hi; u[1] = g.state.lo;
    u[2] = g.inc.hi; u[3] = g.inc.lo;
  }
  return r;
}

// One wavefront: NB column blocks n0, n0 + 16 nstride, ... of dst[16][S] = src[16][S] * Bp^T (both in LDS) TOGETHER: the A
// fragments of a K-tile are read once, the blocks' MFMA chains interleave.  Per block exactly blk_wave_tile's loads,
// staging tile (tb + j * BLK_TB), fragment order and MFMA sequence: the same bits.
static_assert(BLK_PREFETCH == 2, "blk_wave_tiles alternates two register stages");
template <int NB, int PIPE>
__device__ __forceinline__ void blk_wave_tiles(const double *abuf, int S, const double *__restrict__ Bp, int Dp, long long N,
                                               int n0, int nstride, double *out, long long ldo, int lane, double *tb) {
  const int fr = lane & 15, fk = lane >> 4;
  const int r = lane >> 2, kq = lane & 3;
  const int nk = Dp / 16;
  const double *pb[NB];
  d2_t gb[NB][BLK_PREFETCH][2];
  d4_t acc[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    pb[j] = Bp + (long long)(n0 + 16 * nstride * j + r) * Dp + 2 * kq;
    acc[j] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < BLK_PREFETCH; s++) {
      const int kt = s < nk ? s : nk - 1;
      gb[j][s][0] = *reinterpret_cast<const d2_t *>(pb[j] + kt * 16);
      gb[j][s][1] = *reinterpret_cast<const d2_t *>(pb[j] + kt * 16 + 8);
    }
  }
  const double *pa = abuf + fr * S + fk;
  const int xw = (((r >> 1) & 1) << 3) | ((r >> 2) << 1);    // swizzle of this lane's staging row ...
  const int xr = (((fr >> 1) & 1) << 3) | ((fr >> 2) << 1);  // ... and of its fragment row
  // Software pipeline (one wavefront per SIMD: nothing else hides the LDS round trip of the staging tile): while the
  // MFMAs of K-tile kt execute from one set of fragment registers, tile kt + 1 goes registers -> staging tile ->
  // the other set, and the global loads of tile kt + 1 + BLK_PREFETCH are issued.
  auto stage = [&](int s, int kt, bool load) __attribute__((always_inline)) {  // registers of stage s -> staging tiles; refill
#pragma unroll
    for (int j = 0; j < NB; j++) {
      double *t = tb + j * BLK_TB;
      *reinterpret_cast<d2_t *>(&t[r * 16 + ((2 * kq) ^ xw)]) = gb[j][s][0];
      *reinterpret_cast<d2_t *>(&t[r * 16 + ((8 + 2 * kq) ^ xw)]) = gb[j][s][1];
      if (load) {
        const int kn = kt + BLK_PREFETCH < nk ? kt + BLK_PREFETCH : nk - 1;  // (past the end: the last tile again)
        gb[j][s][0] = *reinterpret_cast<const d2_t *>(pb[j] + kn * 16);
        gb[j][s][1] = *reinterpret_cast<const d2_t *>(pb[j] + kn * 16 + 8);
      }
    }
  };
  auto frags = [&](int kt, double (&af)[4], double (&bf)[NB][4]) __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the staging writes: LDS is in order within a wave)
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      af[kk] = pa[kt * 16 + kk * 4];
#pragma unroll
      for (int j = 0; j < NB; j++) bf[j][kk] = tb[j * BLK_TB + fr * 16 + ((kk * 4 + fk) ^ xr)];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the tile is free again)
  };
  auto mfmas = [&](const double (&af)[4], const double (&bf)[NB][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[j][kk], acc[j], 0, 0, 0);
  };
  if (PIPE) {
    double af0[4], bf0[NB][4], af1[4], bf1[NB][4];
    stage(0, 0, true);
    frags(0, af0, bf0);
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {  // tiles kt (set 0) and kt + 1 (set 1); set 0 leaves the loop holding tile kt + 2
      stage(1, kt + 1, true);
      mfmas(af0, bf0);
      frags(kt + 1, af1, bf1);
      stage(0, kt + 2, true);
      mfmas(af1, bf1);
      frags(kt + 2, af0, bf0);
    }
    if (kt + 1 < nk) {  // two tiles left (wave-uniform)
      stage(1, kt + 1, false);
      mfmas(af0, bf0);
      frags(kt + 1, af1, bf1);
      mfmas(af1, bf1);
    } else {
      mfmas(af0, bf0);
    }
  } else {
    double af[4], bf[NB][4];
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
      stage(0, kt, true);
      frags(kt, af, bf);
      mfmas(af, bf);
      stage(1, kt + 1, true);
      frags(kt + 1, af, bf);
      mfmas(af, bf);
    }
    if (kt < nk) {
      stage(0, kt, false);
      frags(kt, af, bf);
      mfmas(af, bf);
    }
  }
  // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const long long col = n0 + 16 * nstride * j + fr;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int mrow = fk + 4 * q;
      if (col < N) out[(long long)mrow * ldo + col] = acc[j][q];
    }
  }
}
// dst[16][S] = src[16][S] * Bp^T over the workgroup's BQ_WAVES wavefronts; the caller places the barriers
template <int PIPE = 1>
__device__ __forceinline__ void blk_gemm_lds4(const double *src, double *dst, int S, const double *Bp, long long D,
                                              int wave, int lane, double *tb) {
  const int NT = (int)((D + 15) / 16);
  const int nb = (NT - wave + BQ_WAVES - 1) / BQ_WAVES;  // this wavefront's blocks: wave, wave + 4, ...
  switch (nb) {
    case 1: blk_wave_tiles<1, PIPE>(src, S, Bp, NT * 16, D, wave * 16, BQ_WAVES, dst, S, lane, tb); break;
    case 2: blk_wave_tiles<2, PIPE>(src, S, Bp, NT * 16, D, wave * 16, BQ_WAVES, dst, S, lane, tb); break;
    case 3: blk_wave_tiles<3, PIPE>(src, S, Bp, NT * 16, D, wave * 16, BQ_WAVES, dst, S, lane, tb); break;
    case 4: blk_wave_tiles<4, PIPE>(src, S, Bp, NT * 16, D, wave * 16, BQ_WAVES, dst, S, lane, tb); break;
    default: break;
  }
}

// R: elements per lane (D <= 16 R)
template <int R, bool TDENSE>
__global__ __launch_bounds__(BQ_THREADS) void k_nuts_block_flow(EngineArgs a, NutsSampleArgs m, BlkFlowArgs f) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tl = lane & 15;                  // lane of the team
  const int ci = wave * 4 + (lane >> 4);     // chain of the workgroup = row of the LDS buffers
  const long long c = (long long)blockIdx.x * BLK_CHAINS + ci;
  const bool valid = c < a.C;
  const long long D = a.D;
  const int S = (int)blk_lds_stride(D);
  double *const xbuf = blk_lds, *const ybuf = blk_lds + BLK_CHAINS * S;
  double *const tb = blk_lds + 2 * BLK_CHAINS * S + wave * (BQ_TILES * BLK_TB);
  double *const mus = blk_lds + 2 * BLK_CHAINS * S + BQ_WAVES * BQ_TILES * BLK_TB;  // the dense target's mean
  double *const park = mus + S + ci * BLK_PARK;  // the chain's generators (sites #2..#4 are drawn from here)
  double *const xrow = xbuf + ci * S, *const yrow = ybuf + ci * S;
  const size_t row = (size_t)(valid ? c : 0) * D;
  for (int k = tl; k < S; k += 16) {  // pads (and the rows of chains past C) stay zero for the whole launch
    xrow[k] = 0.0;
    yrow[k] = 0.0;
  }
  if (TDENSE)
    for (int k = threadIdx.x; k < S; k += BQ_THREADS) mus[k] = k < D ? a.mu[k] : 0.0;
#define EI(r) (tl + 16 * (r))
  // Addresses.  The lock-step engine's work vectors (engine.hip ws_layout: one after the other, the same distance apart)
  // are addressed from ONE base and stride, and the pointers that only once-per-transition code uses (caller state,
  // diagnostics, per-transition records, the launch's momenta) are read from a table in LDS where they are used: with the
  // ~60 pointers of the kernel arguments live in the round loop the scalar register file overflows into VGPR lanes (a
  // first version of this kernel: 1467 v_readlane sites).  vec(k): this lane's element EI(0) of work vector k.
  __shared__ const void *blk_cold[16];
  __shared__ int blk_status[BLK_CHAINS];
  if (threadIdx.x == 0) {
    blk_cold[0] = a.q; blk_cold[1] = a.g; blk_cold[2] = a.U;
    blk_cold[3] = a.out.momentum; blk_cold[4] = a.out.acceptance_probability; blk_cold[5] = a.out.num_doublings;
    blk_cold[6] = a.out.is_turning; blk_cold[7] = a.out.is_diverging; blk_cold[8] = a.out.n_leapfrog;
    blk_cold[9] = m.samples; blk_cold[10] = m.acc_hist; blk_cold[11] = m.div_hist;
    blk_cold[12] = f.p_all; blk_cold[13] = f.v_all; blk_cold[14] = f.w0; blk_cold[15] = m.nleap_total;
  }
#define COLD(T, i) (static_cast<T *>(const_cast<void *>(blk_cold[i])))
  const size_t rowtl = row + tl;
  double *const wsb = a.cur_q;
  const size_t wss = (size_t)(a.cur_p - a.cur_q);
  const int NE = a.max_exp, wmd = 20 + 2 * NE;  // wmd: first vector of the dense-metric group
  auto vec = [&](int k) __attribute__((always_inline)) -> double * { return wsb + ((size_t)k * wss + rowtl); };
  auto cold = [&](int i) __attribute__((always_inline)) -> double * { return COLD(double, i) + rowtl; };
#define EL(ptr, r) (ptr)[16 * (r)]
#define V_END_Q(e) vec(3 + 3 * (e))
#define V_END_P(e) vec(4 + 3 * (e))
#define V_END_G(e) vec(5 + 3 * (e))
#define V_SLOT_Q(s) vec(9 + 3 * (s))
#define V_SLOT_P(s) vec(10 + 3 * (s))
#define V_SLOT_G(s) vec(11 + 3 * (s))
#define V_PSUM vec(15)
#define V_CKP vec(17)
#define V_CKS vec(17 + NE)
#define V_CKV vec(wmd + 3)
#define V_END_V(e) vec(wmd + 1 + (e))
#define V_END_W(e) vec(wmd + 4 + NE + (e))
#define V_SLOT_W(s) vec((s) ? wmd : wmd + 3 + NE)
  const size_t lvl = (size_t)a.C * D;  // checkpoint levels / transitions are C * D doubles apart
  // R = ceil(D / 16) exactly: the slots r < R - 1 are full, only the last one is predicated -- its loads go through a
  // clamped offset (the lane's slot 0 where its last element is past D) and a select, so that no load sits in a branch
  const bool okl = valid && EI(R - 1) < D;
  const int lo = okl ? 16 * (R - 1) : 0;
#define OK(r) ((r) < R - 1 || okl)
#define LDG(ptr, r) ((r) < R - 1 ? (ptr)[16 * (r)] : (okl ? (ptr)[lo] : 0.0))
  // q, p, v and the sub-trajectory momentum sum in registers; dU/dq and w = imm dU/dq stay where the products leave
  // them, in the chain's rows of the two LDS buffers (dense target: P r lands in ybuf, imm g' in xbuf; coordinate-wise
  // target: g' is written to xbuf as the operand, imm g' lands in ybuf).
  // pf: three more vectors that the NEXT bookkeeping step will need -- kind 1: the checkpoint (p, v, momentum sum) of
  // the first U-turn level; kind 2: the other end's p and v and the trajectory's momentum sum.
  // The two proposal slots' w = imm dU/dq live in the moving-end rows cur_w / cur_v, which this kernel does not use.
  double q[R], p[R], v[R], pb[R], pfp[R], pfv[R], pfs[R];
  int pf_kind = 0;
  double *const grow = TDENSE ? yrow : xrow, *const wrow = TDENSE ? xrow : yrow;
#pragma unroll
  for (int r = 0; r < R; r++) {
    q[r] = p[r] = v[r] = pb[r] = pfp[r] = pfv[r] = pfs[r] = 0.0;
  }
  ChainCtl ct = {};
  ct.done = 1;
  double eps = 0.0, U_state = 0.0;
  long long nleap_sum = 0, t = 0;
  bool run = false, pending = false;  // pending: between two transitions (see end_transition)
  bool psum_init = false;  // the trajectory's momentum sum is still the initial momentum (not stored yet)
  bool swapped = false;    // the main proposal has left the initial state
  if (valid) {
    if (tl == 0) {
      const uint64_t *gs = a.rng + (size_t)c * a.nsites * 4;
      unsigned long long *u = reinterpret_cast<unsigned long long *>(park);
      for (int k = 0; k < 4 * a.nsites && k < 16; k++) u[k] = gs[k];
    }
    U_state = a.U[c];
    eps = a.eps_c ? a.eps_c[c] : a.eps;
  }
  BlkTimer tm;

  // first stages of a leapfrog (leap_linear<12>): p_half, v_half, q', the target where it is coordinate-wise, and the
  // operand row of the next product -- r = q' - mu (dense target) or dU/dq' itself -- into this chain's row of xbuf;
  // then the requests for what the bookkeeping behind the products will read (they arrive during the products)
  auto stage12 = [&]() __attribute__((always_inline)) {
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size, aa = 1 * step_size;
    double us[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (OK(r)) {
        const double pp = p[r] - b * grow[EI(r)];
        const double vv = v[r] - b * wrow[EI(r)];
        p[r] = pp;
        v[r] = vv;
        const double qq = q[r] + aa * vv;
        q[r] = qq;
        if (!TDENSE) {
          double u, gnew;
          target_elem(a, EI(r), qq, u, gnew);
          us[r & 3] += u;
          xrow[EI(r)] = gnew;  // (= grow: the new gradient over the old one, and the next product's operand)
        } else {
          xrow[EI(r)] = qq - mus[EI(r)];
        }
      }
    }
    if (!TDENSE) ct.U_cur = target_finish(a, team_sum(us));
    const int step = ct.step;
    if (step & 1) {
      if (pf_kind != 1) {  // (step 1 behind a step 0 that stored under a stale index: level 0 comes from memory)
        const int tmax = __popc(step >> 1);
        const double *kp = V_CKP + (size_t)tmax * lvl, *ks = V_CKS + (size_t)tmax * lvl, *kv = V_CKV + (size_t)tmax * lvl;
#pragma unroll
        for (int r = 0; r < R; r++) {
          pfp[r] = LDG(kp, r);
          pfv[r] = LDG(kv, r);
          pfs[r] = LDG(ks, r);
        }
        pf_kind = 1;
      }
    } else if (step >= 2 && step == (1 << ct.j) && !ct.phantom) {  // the sub-trajectory's last step: expand_once follows
      const int oth = 1 - ct.dir;
      const double *ps = psum_init ? cold(12) + (size_t)t * lvl : V_PSUM;
      const double *ep = V_END_P(oth), *ev = V_END_V(oth);
#pragma unroll
      for (int r = 0; r < R; r++) {
        pfp[r] = LDG(ep, r);
        pfv[r] = LDG(ev, r);
        pfs[r] = LDG(ps, r);
      }
      pf_kind = 2;
    } else {
      pf_kind = 0;
    }
  };
  // sub-trajectory proposal <- moving end (copy_cur_to_slot); w = imm dU/dq travels with it (slot 0: cur_w, slot 1: cur_v)
  auto take = [&](int slot) __attribute__((always_inline)) {
    double *const sq = V_SLOT_Q(slot), *const sp = V_SLOT_P(slot), *const sg = V_SLOT_G(slot), *const sw = V_SLOT_W(slot);
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (OK(r)) {
        EL(sq, r) = q[r];
        EL(sp, r) = p[r];
        EL(sg, r) = grow[EI(r)];
        EL(sw, r) = wrow[EI(r)];
      }
    }
    put2(ct.U_slot, slot, ct.U_cur);
  };
  // expand_once after integrate() returned (nuts_finalize_expansion<true> + nuts_begin_expansion); the outputs of a
  // transition that ends here are written by end_transition / next_transition
  auto finalize = [&](bool is_div, bool has_term) __attribute__((always_inline)) {
    const int dir = ct.dir, oth = 1 - dir;
    if (pf_kind != 2) {  // (an early end -- divergence, sub-tree U-turn -- or the one-step first sub-trajectory)
      const double *ps = psum_init ? cold(12) + (size_t)t * lvl : V_PSUM;
      const double *ep = V_END_P(oth), *ev = V_END_V(oth);
#pragma unroll
      for (int r = 0; r < R; r++) {
        pfp[r] = LDG(ep, r);
        pfv[r] = LDG(ev, r);
        pfs[r] = LDG(ps, r);
      }
    }
    // the state a change of direction continues from, requested before the scalars that decide whether it is needed
    const bool may_go_on = !is_div && !has_term && ct.j + 1 != a.max_exp;
    double eq[R], eg[R], ew[R];
    {
      const double *oq = V_END_Q(oth), *og = V_END_G(oth), *ow = V_END_W(oth);
#pragma unroll
      for (int r = 0; r < R; r++) {
        eq[r] = may_go_on ? LDG(oq, r) : 0.0;
        eg[r] = may_go_on ? LDG(og, r) : 0.0;
        ew[r] = may_go_on ? LDG(ow, r) : 0.0;
      }
    }
    double *const dq = V_END_Q(dir), *const dp = V_END_P(dir), *const dg = V_END_G(dir), *const dv = V_END_V(dir),
                 *const dw = V_END_W(dir), *const dps = V_PSUM;
    double dl[4] = {0.0, 0.0, 0.0, 0.0}, dr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (OK(r)) {
        const double pc = p[r], po = pfp[r], vc = v[r], vo = pfv[r];
        const double s = pfs[r] + pb[r];
        EL(dps, r) = s;
        const double pl = dir ? po : pc, pr = dir ? pc : po;
        const double vl = dir ? vo : vc, vr = dir ? vc : vo;
        const double rho = s - (pr + pl) / 2;
        dl[r & 3] += vl * rho;
        dr[r & 3] += vr * rho;
        EL(dq, r) = q[r];
        EL(dp, r) = pc;
        EL(dg, r) = grow[EI(r)];
        EL(dv, r) = vc;
        EL(dw, r) = wrow[EI(r)];
      }
    }
    psum_init = false;
    const double d_l = team_sum(dl), d_r = team_sum(dr);
    const bool turning = (d_l <= 0) | (d_r <= 0);
    put2(ct.U_end, dir, ct.U_cur);
    // trajectory.py:551-553, proposals.py:130 (always drawn), 141-144, trajectory.py:560-564: the four transcendental
    // chains in four lanes at once (same instruction sequences, same bits: engine.cuh)
    const bool keep = is_div || has_term;
    const ExpansionScalars es = team_expansion_scalars(ct.sub_w, ct.prop_w, ct.sub_slpa, ct.prop_slpa, keep, tl, lane);
    ct.acc_prob = es.e_slpa / (double)ct.length;
    double pbias = es.e_ratio;
    if (pbias > 1.0) pbias = 1.0;
    if (pbias < 0.0) pbias = 0.0;
    const int acc_b = team_bernoulli(park, 3, pbias, tl);
    if (keep) {
      ct.prop_slpa = es.la_slpa;
    } else {
      ct.prop_w = es.la_w;
      ct.prop_slpa = es.la_slpa;
      if (acc_b) {
        ct.prop_slot ^= 1;
        ct.prop_E = ct.sub_E;
        swapped = true;
      }
    }
    ct.ndoubl = ct.j + 1;
    ct.out_div = is_div;
    ct.out_turn = turning;
    const bool end_transition = is_div || turning || has_term || (ct.j + 1 == a.max_exp);
    if (end_transition) {
      ct.done = 1;  // (the caller keeps the chain alive while a phantom scan is pending)
    } else {        // nuts_begin_expansion
      ct.j += 1;
      const int go_right = team_bernoulli(park, 1, 0.5, tl);  // trajectory.py:516
      ct.dir = go_right;
      ct.step = 0;
      if (dir != go_right) {  // cur <- the other end (trajectory.py:518)
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (OK(r)) {
            q[r] = eq[r];
            p[r] = pfp[r];
            grow[EI(r)] = eg[r];
            v[r] = pfv[r];
            wrow[EI(r)] = ew[r];
          }
        }
        ct.U_cur = pick2(ct.U_end, go_right);
      }
    }
  };
  // last stage of the leapfrog + one iteration of dynamic_integration's scan (nuts_book<true, 1>)
  auto book = [&]() __attribute__((always_inline)) {
    const int step = ct.step;
    if (!ct.phantom) ct.nleap += 1;
    int tmin, tmax;
    if (step == 0) {  // termination.py:109-113: indices inherited from the previous sub-trajectory
      tmin = ct.tmin;
      tmax = ct.tmax;
    } else {          // termination.py:192-235 in closed form
      const int n1 = __ffs(~step) - 1;
      tmax = __popc(step >> 1);
      tmin = tmax - n1 + 1;
    }
    const bool even = (step & 1) == 0;
    const bool f_turn = step >= 1 && tmax >= tmin;  // (= step is odd; then pf holds the checkpoint of level tmax)
    const bool deeper = f_turn && tmin < tmax;
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size;
    double *const ckp = V_CKP + (size_t)tmax * lvl;
    double *const cks = V_CKS + (size_t)tmax * lvl;
    double *const ckv = V_CKV + (size_t)tmax * lvl;
    // the second U-turn level, requested before the pass and the scalars (used if the first level does not turn)
    double lp[R], lv[R], ls[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      lp[r] = deeper ? LDG(ckp - lvl, r) : 0.0;
      lv[r] = deeper ? LDG(ckv - lvl, r) : 0.0;
      ls[r] = deeper ? LDG(cks - lvl, r) : 0.0;
    }
    double us[4] = {0.0, 0.0, 0.0, 0.0}, ks[4] = {0.0, 0.0, 0.0, 0.0};
    double fl[4] = {0.0, 0.0, 0.0, 0.0}, fr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (OK(r)) {
        const double gr = grow[EI(r)], wr = wrow[EI(r)];
        if (TDENSE) us[r & 3] += (q[r] - mus[EI(r)]) * gr;  // leap_linear<3>
        const double pn = p[r] - b * gr;
        const double vn = v[r] - b * wr;
        p[r] = pn;
        v[r] = vn;
        ks[r & 3] += vn * pn;                                     // bookkeeping
        const double s = (step == 0) ? pn : pb[r] + pn;
        pb[r] = s;
        if (even) {
          EL(ckp, r) = pn;
          EL(cks, r) = s;
          EL(ckv, r) = vn;
        }
        if (f_turn) {                                             // first level of is_iterative_turning
          const double pl = pfp[r], vl = pfv[r];
          const double sub = s - pfs[r] + pl;
          const double rho = sub - (pn + pl) / 2;
          fl[r & 3] += vl * rho;
          fr[r & 3] += vn * rho;
        }
      }
    }
    tm.tick(0);  // (timing build) vector pass
    if (TDENSE) ct.U_cur = target_finish(a, team_sum(us));
    const double kd = team_sum(ks);
    ct.tmin = tmin;
    ct.tmax = tmax;
    const double E = ct.U_cur + 0.5 * kd;  // proposals.py:19-62
    double delta = ct.H0 - E;
    if (isnan(delta)) delta = -INFINITY;
    const bool div = fabs(delta) > a.thr;
    const double np_w = delta, np_slpa = delta > 0 ? 0.0 : delta;
    bool term = false, do_take = false;
    if (step == 0) {
      ct.sub_E = E;
      ct.sub_w = np_w;
      ct.sub_slpa = np_slpa;
      ct.length = 1;
      do_take = true;
    } else {
      const StepScalars sc = team_step_scalars(ct.sub_w, np_w, ct.sub_slpa, np_slpa, tl, lane);
      const int acc = team_bernoulli(park, 2, sc.pa, tl);
      ct.sub_w = sc.sub_w;
      ct.sub_slpa = sc.sub_slpa;
      if (acc) {
        ct.sub_E = E;
        do_take = !ct.phantom;
      }
      ct.length += 1;
    }
    if (do_take) take(ct.prop_slot ^ 1);  // sub-trajectory proposal <- moving end
    tm.tick(1);  // (timing build) reductions, step scalars, accept draw, proposal copy
    if (f_turn) {  // termination.py:133-187: levels tmax, tmax - 1, ... tmin until one of them turns
      auto dots = [&](const double (&kp_)[R], const double (&kv_)[R], const double (&ks_)[R]) __attribute__((always_inline)) {
        double dl[4] = {0.0, 0.0, 0.0, 0.0}, dr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (OK(r)) {
            const double pl = kp_[r], pr = p[r], vl = kv_[r], vr = v[r];
            const double sub = pb[r] - ks_[r] + pl;
            const double rho = sub - (pr + pl) / 2;
            dl[r & 3] += vl * rho;
            dr[r & 3] += vr * rho;
          }
        }
        const double d_l = team_sum(dl), d_r = team_sum(dr);
        return (d_l <= 0) | (d_r <= 0);
      };
      const double f_dl = team_sum(fl), f_dr = team_sum(fr);  // first level: from the pass above
      bool crit = (f_dl <= 0) | (f_dr <= 0);
      if (!crit && tmax - 1 >= tmin) {
        crit = dots(lp, lv, ls);  // second level: requested at the head of this step
        for (int idx = tmax - 2; !crit && idx >= tmin; idx--) {  // (third level and beyond: steps 7, 15, ...)
          const double *kp = V_CKP + (size_t)idx * lvl, *kss = V_CKS + (size_t)idx * lvl, *kv = V_CKV + (size_t)idx * lvl;
          double xp[R], xv[R], xs[R];
#pragma unroll
          for (int r = 0; r < R; r++) {
            xp[r] = LDG(kp, r);
            xv[r] = LDG(kv, r);
            xs[r] = LDG(kss, r);
          }
          crit = dots(xp, xv, xs);
        }
      }
      term = crit;
    }
    bool fin = false, fin_div = false, fin_term = false, to_phantom = false;
    if (step == 0 && div && !ct.phantom) {
      // trajectory.py:336: integrate() returns the first-step tuple, yet the scan still executes (and draws from
      // site #3): finalize now, keep stepping as a phantom
      fin = fin_div = to_phantom = true;
    } else if (step >= 1 && (div || term || step == (1 << ct.j))) {
      if (ct.phantom) ct.done = 1;
      else {
        fin = true;
        fin_div = div;
        fin_term = term;
      }
    } else {
      ct.step = step + 1;
    }
    // the pair an even step has just stored IS the first U-turn level of the next (odd) step -- unless the step stored
    // under inherited indices (step 0 with tmax != 0: the next step reads level 0 from memory)
    const bool fwd = even && (step != 0 || tmax == 0);
    if (fin) {
      finalize(fin_div, fin_term);
      if (to_phantom) {
        ct.done = 0;
        ct.phantom = 1;
        ct.step = 1;
      }
      pf_kind = 0;  // (pf was consumed or overwritten; the phantom's step 1 fetches its checkpoint)
    } else if (fwd) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        pfp[r] = p[r];
        pfv[r] = v[r];
        pfs[r] = pb[r];
      }
      pf_kind = 1;
    } else {
      pf_kind = 0;
    }
  };
  // nuts_init_chain<true> for transition t: q, p, v in registers, dU/dq (grow) and w (wrow) in their rows
  auto begin_tree = [&]() __attribute__((always_inline)) {
    double ks[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {
      pb[r] = 0.0;
      if (OK(r)) ks[r & 3] += v[r] * p[r];
    }
    const double kd = team_sum(ks);
    const double U = U_state;
    const int s_main = ct.prop_slot;  // the slot that holds the state the transition starts from
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
    ct.prop_slot = s_main;
    ct.ndoubl = ct.out_div = ct.out_turn = 0;
    ct.dir = team_bernoulli(park, 1, 0.5, tl);  // trajectory.py:516
    ct.step = 0;
    psum_init = true;
    swapped = false;
    pf_kind = 0;
    // the trajectory end the first expansion does not move (the other one is written when that expansion ends)
    const int e = 1 - ct.dir;
    double *const dq = V_END_Q(e), *const dp = V_END_P(e), *const dg = V_END_G(e), *const dv = V_END_V(e), *const dw = V_END_W(e);
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (OK(r)) {
        EL(dq, r) = q[r];
        EL(dp, r) = p[r];
        EL(dg, r) = grow[EI(r)];
        EL(dv, r) = v[r];
        EL(dw, r) = wrow[EI(r)];
      }
    }
  };
  // A transition has ended.  Its records now; what comes next -- the accepted state and, unless it was the launch's last
  // transition, the next momentum and velocity -- is only REQUESTED here, into the registers of the finished tree
  // (q <- position, pfp <- dU/dq, pfv <- w, p / v <- next momentum / velocity, or p <- the momentum to report): the
  // chain sits out the next round, the loads arrive during its products, and next_transition() takes over behind them
  // while the running chains do their bookkeeping -- nobody waits for this chain's memory round trips at a barrier.
  auto end_transition = [&]() __attribute__((always_inline)) {
    const int s = ct.prop_slot;
    const long long t_rec = f.t0 + t;
    const bool last = t + 1 == f.nt;
    U_state = pick2(ct.U_slot, s);
    nleap_sum += ct.nleap;
    if (tl == 0) {
      if (COLD(double, 10)) COLD(double, 10)[(size_t)t_rec * a.C + c] = ct.acc_prob;
      if (COLD(int, 11)) COLD(int, 11)[(size_t)t_rec * a.C + c] = ct.out_div;
    }
    // (last transition: the momentum of the accepted proposal -- the initial one if the proposal never changed)
    const double *pn = last ? (swapped ? V_SLOT_P(s) : cold(12) + (size_t)t * lvl) : cold(12) + (size_t)(t + 1) * lvl;
    const double *vn = cold(13) + (size_t)(t + 1) * lvl;
    const double *sq = V_SLOT_Q(s), *sg = V_SLOT_G(s), *sw = V_SLOT_W(s);
#pragma unroll
    for (int r = 0; r < R; r++) {
      q[r] = LDG(sq, r);
      pfp[r] = LDG(sg, r);
      pfv[r] = !last ? LDG(sw, r) : 0.0;
      p[r] = LDG(pn, r);
      v[r] = !last ? LDG(vn, r) : 0.0;
    }
    run = false;
    pending = true;
  };
  auto next_transition = [&]() __attribute__((always_inline)) {
    const long long t_rec = f.t0 + t;
    const bool last = t + 1 == f.nt;
    pending = false;
    if (COLD(double, 9)) {
      double *dst = cold(9) + (size_t)t_rec * lvl;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (OK(r)) EL(dst, r) = q[r];
    }
    if (last) {  // nuts_write_outputs
      double *const oq = cold(0), *const og = cold(1), *const om = COLD(double, 3) ? cold(3) : nullptr;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (OK(r)) {
          EL(oq, r) = q[r];
          EL(og, r) = pfp[r];
          if (om) EL(om, r) = p[r];
        }
      }
      if (tl == 0) {
        COLD(double, 2)[c] = U_state;
        COLD(double, 4)[c] = ct.acc_prob;
        if (COLD(int64_t, 5)) COLD(int64_t, 5)[c] = ct.ndoubl;
        if (COLD(int32_t, 6)) COLD(int32_t, 6)[c] = ct.out_turn;
        COLD(int32_t, 7)[c] = ct.out_div;
        if (COLD(int64_t, 8)) COLD(int64_t, 8)[c] = ct.nleap;
      }
    } else {
      t += 1;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (OK(r)) {
          grow[EI(r)] = pfp[r];
          wrow[EI(r)] = pfv[r];
        }
      }
      begin_tree();
      run = true;
    }
  };

  __syncthreads();  // (the zeroed rows, the mean, the generators)
  if (valid && f.nt > 0) {  // the launch's first transition: the caller's state, w0 = imm dU/dq, both into proposal slot 0
    const double *iq = cold(0), *ig = cold(1), *ip = cold(12), *iv = cold(13), *iw = cold(14);
    double *const sq = V_SLOT_Q(0), *const sg = V_SLOT_G(0), *const sw = V_SLOT_W(0);
#pragma unroll
    for (int r = 0; r < R; r++) {
      q[r] = LDG(iq, r);
      p[r] = LDG(ip, r);
      v[r] = LDG(iv, r);
      if (OK(r)) {
        const double g0 = EL(ig, r), w0 = EL(iw, r);
        grow[EI(r)] = g0;
        wrow[EI(r)] = w0;
        EL(sq, r) = q[r];
        EL(sg, r) = g0;
        EL(sw, r) = w0;
      }
    }
    ct.prop_slot = 0;
    begin_tree();
    run = true;
  }
  tm.tick(7);
  // ---- one round: a leapfrog of every running chain -- first stages | P r | imm g' | last stage + bookkeeping ----
  for (;;) {
    if (run) stage12();
    tm.tick(5);
    if (tl == 0) blk_status[ci] = (run || pending) ? 1 : 0;
    blk_barrier_lds();
    int live = 0;
#pragma unroll
    for (int k = 0; k < BLK_CHAINS; k++) live |= blk_status[k];
    live = __builtin_amdgcn_readfirstlane(live);
    tm.tick(6);  // vote (waits for the slowest wavefront's bookkeeping)
    if (!live) break;
    if (TDENSE) {
      blk_gemm_lds4(xbuf, ybuf, S, m.prec, D, wave, lane, tb);  // dU/dq' = P r
      tm.tick(2);
      blk_barrier_lds();
      tm.tick(3);
      blk_gemm_lds4(ybuf, xbuf, S, a.imm, D, wave, lane, tb);   // w' = imm dU/dq'
      tm.tick(2);
      blk_barrier_lds();
      tm.tick(3);
    } else {
      blk_gemm_lds4(xbuf, ybuf, S, a.imm, D, wave, lane, tb);   // w' = imm dU/dq'
      tm.tick(2);
      blk_barrier_lds();
      tm.tick(3);
    }
    if (run) {
      book();
      if (ct.done) end_transition();
    } else if (pending) {
      next_transition();
    }
    tm.tick(4);
  }
  if (valid) {
    if (tl == 0) {  // (site #1 was advanced by the momentum pre-pass)
      uint64_t *gs = a.rng + (size_t)c * a.nsites * 4;
      const unsigned long long *u = reinterpret_cast<const unsigned long long *>(park);
      for (int k = 4; k < 4 * a.nsites && k < 16; k++) gs[k] = u[k];
      if (COLD(long long, 15)) COLD(long long, 15)[c] += nleap_sum;
    }
  }
#ifdef AEHMC_WIDE_TIMING
  if (lane == 0)
    for (int k = 0; k < 8; k++) a.linreg_part[((size_t)blockIdx.x * BQ_WAVES + wave) * 8 + k] = (double)tm.acc[k];
#endif
#undef EI
#undef EL
#undef OK
#undef LDG
#undef COLD
#undef V_END_Q
#undef V_END_P
#undef V_END_G
#undef V_SLOT_Q
#undef V_SLOT_P
#undef V_SLOT_G
#undef V_PSUM
#undef V_CKP
#undef V_CKS
#undef V_CKV
#undef V_END_V
#undef V_END_W
#undef V_SLOT_W
}

template <int R>
inline hipError_t launch_nuts_block_flow_r(const EngineArgs &a, const NutsSampleArgs &m, const BlkFlowArgs &f,
                                           hipStream_t st) {
  const size_t dyn = blk_flow_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BQ_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_block_flow<R, TDV>),           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_nuts_block_flow<R, TDV>), grid, block, dyn, st, a, m, f);                          \
  } while (0)
  if (a.tkind == AEHMC_T_DENSE_MVN) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}
// k_nuts_block_flow addresses the work vectors as cur_q + k * (cur_p - cur_q) (engine.hip ws_layout)
inline bool blk_flow_layout_ok(const EngineArgs &a) {
  const ptrdiff_t s = a.cur_p - a.cur_q;
  const ptrdiff_t E = a.max_exp, md = 20 + 2 * E;
  auto at = [&](const double *p, ptrdiff_t k) { return p == a.cur_q + k * s; };
  bool ok = s > 0 && at(a.psum, 15) && at(a.ckp, 17) && at(a.cks, 17 + E) && at(a.cur_v, md) && at(a.ckv, md + 3) &&
            at(a.cur_w, md + 3 + E);
  for (int e = 0; e < 2; e++)
    ok = ok && at(a.end_q[e], 3 + 3 * e) && at(a.end_p[e], 4 + 3 * e) && at(a.end_g[e], 5 + 3 * e) &&
         at(a.slot_q[e], 9 + 3 * e) && at(a.slot_p[e], 10 + 3 * e) && at(a.slot_g[e], 11 + 3 * e) &&
         at(a.end_v[e], md + 1 + e) && at(a.end_w[e], md + 4 + E + e);
  return ok;
}
// bp: the launch's packed matrices (blk_pack_matrices); f: the momenta, velocities and w0 formed by the caller
inline hipError_t launch_nuts_block_flow(EngineArgs a, NutsSampleArgs m, const BlkFlowArgs &f, double *bp, hipStream_t st) {
  if (!blk_flow_layout_ok(a)) return hipErrorInvalidValue;
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, m.prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass; m.prec = mats.prec;
  switch ((int)((a.D + 15) / 16)) {  // elements per lane, exactly
    case 5: return launch_nuts_block_flow_r<5>(a, m, f, st);
    case 6: return launch_nuts_block_flow_r<6>(a, m, f, st);
    case 7: return launch_nuts_block_flow_r<7>(a, m, f, st);
    case 8: return launch_nuts_block_flow_r<8>(a, m, f, st);
    case 9: return launch_nuts_block_flow_r<9>(a, m, f, st);
    case 10: return launch_nuts_block_flow_r<10>(a, m, f, st);
    case 11: return launch_nuts_block_flow_r<11>(a, m, f, st);
    case 12: return launch_nuts_block_flow_r<12>(a, m, f, st);
    case 13: return launch_nuts_block_flow_r<13>(a, m, f, st);
    case 14: return launch_nuts_block_flow_r<14>(a, m, f, st);
    case 15: return launch_nuts_block_flow_r<15>(a, m, f, st);
    case 16: return launch_nuts_block_flow_r<16>(a, m, f, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace aehmc
