// Block-resident dense NUTS / HMC (gfx950): mid-size dense problems, 64 < D <= 512, in ONE launch.
//
// A dense inverse mass matrix (shared by the chains) and / or the dense-precision target make the leapfrog's
// metric / gradient products real matrix products (metrics.py:52-73,94-102).  Up to D = 64 a wavefront does them
// itself (nuts_resident.cuh, DENSE); for large D the lock-step engine batches ALL chains into one fp64 MFMA GEMM
// per product (engine.cuh + gemm_f64.cuh: 73 TFLOP/s at D = 1e4).  In between -- D = 100 ... 500 with a full mass
// matrix, the everyday use of this path -- a lock-step is four launches of 10-20 us each around products that are too
// small to fill the GPU (53 us per lock-step at D = 200, 4096 chains: round 3), every finished chain waits for the
// deepest tree of all chains, and the host polls.
//
// Here ONE workgroup of 16 wavefronts owns 16 chains for the whole launch (no grid sync, no live-chain compaction,
// no host poll):
//   * stages and NUTS bookkeeping: wavefront w owns chain 16 b + w and runs the lock-step engine's own device
//     functions on it (leap_linear, nuts_book, nuts_init_chain ... engine.cuh) -- the chain's vectors are rows of
//     the same work arrays, L2-resident (16 chains x ~25 vectors x 8 D bytes per workgroup);
//   * products: the 16 chains are the 16 rows of one MFMA tile.  Per product the operand rows are staged into LDS
//     ([16][D16 + 2] doubles, conflict-free fragment reads) and every wavefront computes 16 x 16 output blocks
//     with v_mfma_f64_16x16x4_f64, streaming its 16 rows of the matrix from L2 in full 128-byte lines four K-tiles
//     ahead; the loaded 4 x 4 register blocks are brought into fragment order with v_permlane32_swap /
//     v_permlane16_swap (gfx950), so the matrix never passes through LDS.  Linear dense mode: two products per
//     leapfrog (P r and imm g', engine.cuh leap_linear), three more at the start of a transition (L^-T z, imm p, imm g).
//   * every output element is the k-chain of the chain-batched GEMM kernels -- accumulator from 0, K-tiles of 16
//     in ascending k, four MFMAs per tile with k = k0 + 4 kk + (lane >> 4) -- so the results are BITWISE those of
//     the lock-step path (tests/test_gpu_block_dense.py), whichever chains share a workgroup.
//   * kernel.sample(T): the 16 chains of a workgroup start each transition together; workgroups are independent,
//     so a transition costs the deepest tree among 16 chains, not among all of them.
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235, hmc.py:77-204,
// integrators.py:54-73, metrics.py:44-104.
#pragma once
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime itself) */
#include <hip/hip_runtime.h>
#endif

#include "engine.cuh"
#ifndef __HIPCC_RTC__
#include "gemm_f64.cuh"
#else  // run-time compiled copy (user-defined target with a dense metric): only the vector types of gemm_f64.cuh
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
#endif

namespace aehmc {

constexpr int BLK_CHAINS = 16;                  // chains per workgroup = rows of one MFMA tile
constexpr int BLK_THREADS = 64 * BLK_CHAINS;    // one wavefront per chain for the stages
constexpr int BLK_MIN_D = 65, BLK_MAX_D = 512;  // below: inside the wavefront; above: chain-batched GEMM
#ifndef AEHMC_BLK_PREFETCH
#define AEHMC_BLK_PREFETCH 2
#endif
constexpr int BLK_PREFETCH = AEHMC_BLK_PREFETCH;  // K-tiles of the matrix in flight per wavefront

__host__ __device__ inline long long blk_lds_stride(long long D) { return (D + 15) / 16 * 16 + 2; }  // = 2 mod 4: fragment reads hit every bank twice
constexpr int BLK_TB = 16 * 16;  // doubles of a wavefront's B staging tile ([16 rows][16], columns swizzled)
// operand rows [16][S] + one staging tile per wavefront
#ifndef __HIPCC_RTC__  // (host side)
inline size_t blk_lds_bytes(long long D) {
  return ((size_t)BLK_CHAINS * blk_lds_stride(D) + (size_t)BLK_CHAINS * BLK_TB) * sizeof(double);
}

inline bool block_dense_supported(int tkind, int met_ndim, int per_chain, long long D) {
  const bool elem = tkind == AEHMC_T_STD_NORMAL || tkind == AEHMC_T_ISO_GAUSSIAN || tkind == AEHMC_T_DIAG_GAUSSIAN;
  return met_ndim == 2 && !per_chain && (elem || tkind == AEHMC_T_DENSE_MVN) && D >= BLK_MIN_D && D <= BLK_MAX_D;
}
#endif  // __HIPCC_RTC__

// The matrices of a launch, zero-padded to [Dp][Dp] with Dp = D rounded up to 16 (a ctx-owned copy, rewritten by
// every call: the caller's matrices may have changed): the product loop then needs no bounds predicates at all --
// with predicated loads the compiler waits for ALL outstanding loads before each use (s_waitcnt vmcnt(0)) and the
// prefetch depth collapses to one tile -- every 16-byte load is aligned whatever D, and rows / columns past D
// contribute exact zeros.
AEHMC_TU_LOCAL __global__ __launch_bounds__(256) void k_blk_pack(const double *src, double *dst, long long D, long long Dp) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Dp * Dp) return;
  const long long i = e / Dp, j = e % Dp;
  dst[e] = (i < D && j < D) ? src[i * D + j] : 0.0;
}

// One wavefront: out[m][n0 + n] = sum_k A[m][k] B[n0 + n][k] for the 16 chains m of the workgroup and 16 columns n
// -- the A rows in LDS (abuf [16][S], zero beyond D), B packed row-major [Dp][Dp] in global memory (L2), the result
// to out[m * ldo + n0 + n] for the rows m of `rowmask` (bit m), n0 + n < N (global rows of the work arrays, or rows of
// an LDS buffer; rows are independent, so a masked-out row costs nothing but its share of the MFMA).
// B is fetched in whole cache lines: lane (r = lane >> 2, kq = lane & 3) loads doubles 2 kq, 2 kq + 1 and 8 + 2 kq,
// 9 + 2 kq of row n0 + r of the K-tile -- four consecutive lanes read 64 contiguous bytes (a first version had lane
// (row = lane & 15, quarter = lane >> 4) load its fragment's 32 bytes directly: 64 separate 16-byte requests per
// load instruction, and the CU's L1 tag rate, not the MFMA pipe, bounded the product at 38 % of peak) --
// BLK_PREFETCH tiles ahead in registers, and passes through a wave-private LDS tile (tb [16][16]) into MFMA fragment
// order: lane (fr = lane & 15, fk = lane >> 4) holds B[n0 + fr][k0 + 4 kk + fk] for the tile's four MFMAs kk.
// The tile's columns are swizzled -- column k of row r sits at k ^ x(r >> 1), x(t) = ((t & 1) << 3) | ((t >> 1) << 1)
// -- so that the 16-byte staging writes of 16 consecutive lanes (4 rows x 4 chunks) AND the 8-byte fragment reads
// of 16 consecutive lanes (16 rows, one column) each cover all 64 banks once.  Measured (tools/debug/
// blk_gemm_ablate.hip, D = 512): MFMAs alone 64.6 cycles each per SIMD, + A fragments from LDS 66, + this tile padded
// to 18 doubles per row (conflict-free reads, two-way conflicts on the writes) 98, swizzled 77; the global loads
// and the wavefront fences add nothing.
// The first BLK_PREFETCH K-tiles of a wavefront's column block of B, requested AHEAD of the product (round 6): B does not
// depend on anything the chains compute, so the block-resident kernels issue these loads before the workgroup barrier
// that precedes the product -- the L2 round trip (~1 000 cycles at the head of every product, with the MFMA pipe idle)
// passes while the wavefront waits at the barrier.
struct BlkPre {
  d2_t gb[BLK_PREFETCH][2];
};
__device__ __forceinline__ void blk_tile_prefetch(const double *__restrict__ Bp, int Dp, int n0, int lane, BlkPre &pre) {
  const int r = lane >> 2, kq = lane & 3;
  const double *pb = Bp + (long long)(n0 + r) * Dp + 2 * kq;
  const int nk = Dp / 16;
#pragma unroll
  for (int s = 0; s < BLK_PREFETCH; s++) {
    const int kt = s < nk ? s : nk - 1;
    pre.gb[s][0] = *reinterpret_cast<const d2_t *>(pb + kt * 16);
    pre.gb[s][1] = *reinterpret_cast<const d2_t *>(pb + kt * 16 + 8);
  }
}
template <bool PRE = false>
__device__ __forceinline__ void blk_wave_tile(const double *abuf, int S, const double *__restrict__ Bp, int Dp,
                                              long long N, int n0, double *out, long long ldo, unsigned rowmask,
                                              int lane, double *tb, const BlkPre *pre = nullptr) {
  const int fr = lane & 15, fk = lane >> 4;
  const int r = lane >> 2, kq = lane & 3;
  const double *pb = Bp + (long long)(n0 + r) * Dp + 2 * kq;
  const int nk = Dp / 16;
  d2_t gb[BLK_PREFETCH][2];
  d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < BLK_PREFETCH; s++) {
    if (PRE) {
      gb[s][0] = pre->gb[s][0];
      gb[s][1] = pre->gb[s][1];
    } else {
      const int kt = s < nk ? s : nk - 1;
      gb[s][0] = *reinterpret_cast<const d2_t *>(pb + kt * 16);
      gb[s][1] = *reinterpret_cast<const d2_t *>(pb + kt * 16 + 8);
    }
  }
  const double *pa = abuf + fr * S + fk;
  const int xw = (((r >> 1) & 1) << 3) | ((r >> 2) << 1);    // swizzle of this lane's staging row ...
  const int xr = (((fr >> 1) & 1) << 3) | ((fr >> 2) << 1);  // ... and of its fragment row
  // one K-tile: registers of stage s -> staging tile -> fragments -> four MFMAs; LOAD: refill the stage
  auto tile = [&](int s, int kt, bool load) __attribute__((always_inline)) {
    *reinterpret_cast<d2_t *>(&tb[r * 16 + ((2 * kq) ^ xw)]) = gb[s][0];
    *reinterpret_cast<d2_t *>(&tb[r * 16 + ((8 + 2 * kq) ^ xw)]) = gb[s][1];
    if (load) {
      const int kn = kt + BLK_PREFETCH < nk ? kt + BLK_PREFETCH : nk - 1;  // (past the end: the last tile again)
      gb[s][0] = *reinterpret_cast<const d2_t *>(pb + kn * 16);
      gb[s][1] = *reinterpret_cast<const d2_t *>(pb + kn * 16 + 8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS is in order within a wave
    double af[4], bf[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      af[kk] = pa[kt * 16 + kk * 4];
      bf[kk] = tb[fr * 16 + ((kk * 4 + fk) ^ xr)];
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], acc, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  // main loop: whole groups of BLK_PREFETCH tiles, no exit inside a group (with one, the compiler's wait counts at the
  // head of the loop fall back to "all loads but one": the prefetch depth collapsed to one tile there); then the rest
  int kt0 = 0;
  for (; kt0 + BLK_PREFETCH <= nk; kt0 += BLK_PREFETCH) {
#pragma unroll
    for (int s = 0; s < BLK_PREFETCH; s++) tile(s, kt0 + s, true);
  }
#pragma unroll
  for (int s = 0; s < BLK_PREFETCH - 1; s++)
    if (kt0 + s < nk) tile(s, kt0 + s, false);  // wave-uniform
  // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const long long col = n0 + fr;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int mrow = fk + 4 * q;
    if (((rowmask >> mrow) & 1u) && col < N) out[(long long)mrow * ldo + col] = acc[q];
  }
}

// Developer instrumentation (make timing): shader-clock cycles per phase, accumulated by every wavefront and written
// to a.linreg_part[c * 16 + phase] (unused workspace on this path); compiled out of the product library.
struct BlkTimer {
#ifdef AEHMC_WIDE_TIMING
  long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long last = (long long)__builtin_amdgcn_s_memtime();
  __device__ __forceinline__ void tick(int k) {
    const long long now = (long long)__builtin_amdgcn_s_memtime();
    acc[k] += now - last;
    last = now;
  }
#else
  __device__ __forceinline__ void tick(int) {}
#endif
};

// out[c0 + m][:] = X[c0 + m][:] * B^T for the workgroup's 16 chains (all [.,D] row-major, D x D matrix B).  Wavefront w
// stages row w of X into LDS (the row it wrote itself in the stage before, or -- between two products -- rows
// written by other wavefronts before the previous barrier), then the wavefronts share the D / 16 column blocks.
// Ends with a barrier: `out` is visible to the whole workgroup and the LDS rows are free again.
__device__ __forceinline__ void blk_gemm(double *abuf, int S, const double *X, const double *B, double *out, long long D,
                                         long long c0, long long C, int wave, int lane, BlkTimer &tm) {
  const long long c = c0 + wave;
  for (int k = lane; k < S; k += 64) abuf[wave * S + k] = (c < C && k < D) ? X[c * D + k] : 0.0;
  tm.tick(0);  // operand rows -> LDS
  __syncthreads();
  tm.tick(1);  // barrier
  const int NT = (int)((D + 15) / 16);
  const int mvalid = (int)(C - c0 < BLK_CHAINS ? C - c0 : BLK_CHAINS);
  double *const tb = abuf + BLK_CHAINS * S + wave * BLK_TB;  // this wavefront's staging tile
  for (int nt = wave; nt < NT; nt += BLK_CHAINS)
    blk_wave_tile(abuf, S, B, NT * 16, D, nt * 16, out + c0 * D, D, (1u << mvalid) - 1u, lane, tb);
  tm.tick(2);  // MFMA column blocks
  __syncthreads();
  tm.tick(3);  // barrier
}

// NUTS: nuts_run's lock-step loop (engine.hip) for 16 chains, in one launch.  Same device functions, same products
// (linear dense mode), hence the same bits.  m.T transitions per launch (kernel.sample); per-transition records optional.
template <bool TDENSE>
__global__ __launch_bounds__(BLK_THREADS) void k_nuts_block_dense(EngineArgs a, NutsSampleArgs m) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS, c = c0 + wave;
  const bool valid = c < a.C;
  const int S = (int)blk_lds_stride(a.D);
  const size_t row = (size_t)(valid ? c : 0) * a.D;
  ChainRng rng = {};
  ChainCtl ct = {};
  ct.done = 1;
  double U_state = 0.0;
  long long nleap_sum = 0;
  BlkTimer tm;
  if (valid) {
    rng = rng_load(a, c);
    U_state = a.U[c];
  }
  for (long long t_idx = 0; t_idx < m.T; t_idx++) {
    // ---- momentum: p = L^-T z (metrics.py:65-68), v = imm p, w = imm g (nuts.py:113-125) ----
    if (valid) draw_momentum<true>(a, c, lane, rng.g[0]);
    tm.tick(7);
    blk_gemm(blk_lds, S, a.zbuf, a.sqrt_mass, a.cur_p, a.D, c0, a.C, wave, lane, tm);
    blk_gemm(blk_lds, S, a.cur_p, a.imm, a.cur_v, a.D, c0, a.C, wave, lane, tm);
    blk_gemm(blk_lds, S, a.g, a.imm, a.cur_w, a.D, c0, a.C, wave, lane, tm);
    if (valid) {
      nuts_init_chain<true>(a, c, lane, ct, rng, &U_state);
      double U_next = 0.0;
      if (leap_linear<12>(a, c, lane, ct.dir, U_next)) ct.U_cur = U_next;
    }
    tm.tick(7);
    // ---- one leapfrog of every live chain per trip: P r | imm g' | last stage + bookkeeping + next first stages ----
    for (;;) {
      const int live = __syncthreads_or(valid && !ct.done);
      tm.tick(6);  // "any chain alive" vote (waits for the slowest stage)
      if (!live) break;
      if (TDENSE) blk_gemm(blk_lds, S, a.rbuf, m.prec, a.cur_g, a.D, c0, a.C, wave, lane, tm);
      blk_gemm(blk_lds, S, a.cur_g, a.imm, a.cur_w, a.D, c0, a.C, wave, lane, tm);
      if (valid && !ct.done) {
        nuts_book<true, 1>(a, c, lane, ct, rng);
        tm.tick(4);  // last stage + bookkeeping
        if (!ct.done) {
          __threadfence_block();
          double U_next = 0.0;
          if (leap_linear<12>(a, c, lane, ct.dir, U_next)) ct.U_cur = U_next;
        }
        tm.tick(5);  // first stages of the next leapfrog
      }
    }
    // ---- per-transition records (the outputs themselves were written by nuts_write_outputs) ----
    if (valid) {
      U_state = pick2(ct.U_slot, ct.prop_slot);
      nleap_sum += ct.nleap;
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
  if (valid) {
    rng_store(a, c, lane, rng, 0, 3);
    if (lane == 0 && m.nleap_total) m.nleap_total[c] = nleap_sum;
#ifdef AEHMC_WIDE_TIMING
    if (lane == 0)
      for (int k = 0; k < 8; k++) a.linreg_part[c * 8 + k] = (double)tm.acc[k];
#endif
  }
}

// HMC: hmc_run's lock-step loop (engine.hip) for 16 chains, nt transitions x L leapfrogs in one launch.
template <bool TDENSE>
__global__ __launch_bounds__(BLK_THREADS) void k_hmc_block_dense(EngineArgs a, const double *prec, long long L,
                                                                  long long nt, double *samples, double *acc_hist,
                                                                  int *div_hist) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS, c = c0 + wave;
  const bool valid = c < a.C;
  const int S = (int)blk_lds_stride(a.D);
  const size_t row = (size_t)(valid ? c : 0) * a.D;
  Pcg64 g1 = {}, g2 = {};
  double U_state = 0.0;
  BlkTimer tm;
  if (valid) {
    g1 = pcg_load(a.rng + (size_t)c * a.nsites * 4);
    g2 = pcg_load(a.rng + ((size_t)c * a.nsites + 1) * 4);
    U_state = a.U[c];
  }
  for (long long tt = 0; tt < nt; tt++) {
    if (valid) draw_momentum<true>(a, c, lane, g1);
    blk_gemm(blk_lds, S, a.zbuf, a.sqrt_mass, a.cur_p, a.D, c0, a.C, wave, lane, tm);
    blk_gemm(blk_lds, S, a.cur_p, a.imm, a.cur_v, a.D, c0, a.C, wave, lane, tm);
    blk_gemm(blk_lds, S, a.g, a.imm, a.cur_w, a.D, c0, a.C, wave, lane, tm);
    ChainCtl ct = {};
    if (valid) ct = hmc_init_chain<true>(a, c, lane, &U_state);
    for (long long l = 0; l < L; l++) {  // trajectory.py:86-95
      if (valid) {
        double U_new = 0.0;
        if (leap_linear<12>(a, c, lane, 1, U_new)) ct.U_cur = U_new;
      }
      if (TDENSE) blk_gemm(blk_lds, S, a.rbuf, prec, a.cur_g, a.D, c0, a.C, wave, lane, tm);
      blk_gemm(blk_lds, S, a.cur_g, a.imm, a.cur_w, a.D, c0, a.C, wave, lane, tm);
      if (valid) {
        double U_new = 0.0;
        if (leap_linear<3>(a, c, lane, 1, U_new)) ct.U_cur = U_new;
      }
    }
    if (valid) {
      __threadfence_block();
      const HmcEnd e = hmc_end_chain_rng<true>(a, c, lane, ct, L, g2);
      if (e.acc) U_state = ct.U_cur;
      if (samples) {  // (a.q: what this lane has just written, or left untouched on rejection)
        double *dst = samples + ((size_t)tt * a.C + c) * a.D;
        for (long long i = lane; i < a.D; i += 64) dst[i] = a.q[row + i];
      }
      if (lane == 0) {
        if (acc_hist) acc_hist[(size_t)tt * a.C + c] = e.pa;
        if (div_hist) div_hist[(size_t)tt * a.C + c] = e.is_div;
      }
    }
  }
  if (valid && lane == 0) {
    pcg_store(a.rng + (size_t)c * a.nsites * 4, g1);
    pcg_store(a.rng + ((size_t)c * a.nsites + 1) * 4, g2);
  }
}

#ifndef __HIPCC_RTC__  // (host side)
// bp: the packed matrices [3][Dp][Dp] -- inverse mass matrix, L^-T, precision (dense target only)
struct BlkMats {
  const double *imm, *sqrt_mass, *prec;
};
inline hipError_t blk_pack_matrices(const EngineArgs &a, const double *prec, double *bp, BlkMats &mats, hipStream_t st) {
  const long long Dp = (a.D + 15) / 16 * 16;
  const unsigned grid = (unsigned)((Dp * Dp + 255) / 256);
  hipLaunchKernelGGL(k_blk_pack, dim3(grid), dim3(256), 0, st, a.imm, bp, (long long)a.D, Dp);
  hipLaunchKernelGGL(k_blk_pack, dim3(grid), dim3(256), 0, st, a.sqrt_mass, bp + Dp * Dp, (long long)a.D, Dp);
  if (a.tkind == AEHMC_T_DENSE_MVN)
    hipLaunchKernelGGL(k_blk_pack, dim3(grid), dim3(256), 0, st, prec, bp + 2 * Dp * Dp, (long long)a.D, Dp);
  mats.imm = bp;
  mats.sqrt_mass = bp + Dp * Dp;
  mats.prec = bp + 2 * Dp * Dp;
  return hipGetLastError();
}
inline size_t blk_pack_bytes(long long D) {
  const long long Dp = (D + 15) / 16 * 16;
  return (size_t)3 * Dp * Dp * sizeof(double);
}

inline hipError_t launch_nuts_block_dense(EngineArgs a, NutsSampleArgs m, double *bp, hipStream_t st) {
  const bool td = a.tkind == AEHMC_T_DENSE_MVN;
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, m.prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass; m.prec = mats.prec;
  const size_t dyn = blk_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BLK_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_block_dense<TDV>),             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_nuts_block_dense<TDV>), grid, block, dyn, st, a, m);                               \
  } while (0)
  if (td) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}

inline hipError_t launch_hmc_block_dense(EngineArgs a, const double *prec, long long L, long long nt,
                                         double *samples, double *acc_hist, int *div_hist, double *bp, hipStream_t st) {
  const bool td = a.tkind == AEHMC_T_DENSE_MVN;
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass;
  const size_t dyn = blk_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BLK_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hmc_block_dense<TDV>),              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_hmc_block_dense<TDV>), grid, block, dyn, st, a, mats.prec, L, nt, samples, acc_hist, \
                       div_hist);                                                                            \
  } while (0)
  if (td) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}
#endif  // __HIPCC_RTC__

}  // namespace aehmc
