// fp64 MFMA GEMM for the dense-metric / dense-precision path (gfx950).
//
//   Cmat[M,N] = A[M,K] * B[N,K]^T        (all row-major, K contiguous in both operands)
//
// This is the "p <- M^-1 p is a real GEMM" case of the north star: with a dense inverse
// mass matrix the reference's per-chain gemv `at.dot(inverse_mass_matrix, momentum)`
// (aehmc/metrics.py:71,95-96) and the dense target gradient become, over the chains of
// one lock-step leapfrog, one [C,D] x [D,D] product.  Rows = chains, so every output
// element's k-summation order is independent of how many chains are batched.
//
// Tiling (v_mfma_f64_16x16x4_f64, wave64), BK = 16, LDS rows padded to 18 doubles, two LDS
// stages, one barrier per K-tile, 256 threads = 4 waves in a 2x2 arrangement:
//   gemm_nt_f64_streamk_kernel<.., 8>  128x256 tile, one workgroup per CU, each wave 64x128 =
//       4x8 MFMA tiles (128 f64 accumulators / lane, in AGPRs), software-pipelined K loop --
//       the default for the chain-batched products (73 TFLOP/s at 4096 x 1e4 x 1e4);
//   gemm_nt_f64_streamk_kernel<.., 4>  128x128 tile, two workgroups per CU, compiler-scheduled;
//   gemm_nt_f64_kernel                 one 128x128 tile per workgroup (few tiles; modes for the
//       blocked Cholesky / triangular inverse).
// Block ids are dealt to XCDs round-robin by the hardware, so each XCD is given a contiguous
// run of tiles that share operand panels in its L2.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace aehmc {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16, GEMM_LDS = GEMM_BK + 2;
constexpr int GEMM_PANEL_W = 8;  // 64 concurrent tiles of an XCD = 8 x 8 super-tile

struct GemmTileMap {
  int tm, tn;
  bool valid;
};
// block id -> tile; see header comment
__device__ __forceinline__ GemmTileMap gemm_tile_of_block(int b, int Tm, int Tn) {
  const int total = Tm * Tn;
  const int chunk = (total + 7) / 8;
  const int t = (b % 8) * chunk + (b / 8);
  GemmTileMap r;
  r.valid = (b / 8) < chunk && t < total;
  const int per_panel = Tm * GEMM_PANEL_W;
  const int nfull = Tn / GEMM_PANEL_W;
  int panel = t / per_panel;
  if (panel < nfull) {
    int rr = t - panel * per_panel;
    r.tn = panel * GEMM_PANEL_W + rr % GEMM_PANEL_W;
    r.tm = rr / GEMM_PANEL_W;
  } else {
    int wlast = Tn - nfull * GEMM_PANEL_W;  // 1 when Tn is odd
    int rr = t - nfull * per_panel;
    if (wlast < 1) wlast = 1;
    r.tn = nfull * GEMM_PANEL_W + rr % wlast;
    r.tm = rr / wlast;
  }
  return r;
}

// loads this thread's 8 doubles of a 128 x 16 operand tile (row = tid/2, cols (tid&1)*8..+7);
// `src_row` is the memory row of tile row tid/2 (or < 0 when that row is out of range)
template <bool VEC>
__device__ __forceinline__ void gemm_load_tile(const double *__restrict__ P, int64_t ld,
                                               int64_t src_row, int64_t k0, int64_t K, int tid,
                                               double (&r)[8]) {
  const int64_t k = k0 + (tid & 1) * 8;
  if (src_row >= 0) {
    const double *src = P + src_row * ld + k;
    if (VEC && k + 8 <= K) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        d2_t v = *reinterpret_cast<const d2_t *>(src + 2 * i);
        r[2 * i] = v[0];
        r[2 * i + 1] = v[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) r[i] = (k + i < K) ? src[i] : 0.0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = 0.0;
  }
}

// row_idx / n_rows (both optional): compacted-row mode for the lock-step engine -- tile
// row r reads A row row_idx[r] and writes C row row_idx[r], and only *n_rows rows exist
// (chains whose transition has finished drop out of the product).
// MODE 0: C = A B^T;  1: C = C - A B^T;  2: C = -A B^T  (1, 2: the blocked Cholesky /
// triangular inverse of aehmc_set_metric)
template <bool VEC, int MODE = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(
    int64_t M, int64_t N, int64_t K, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double *__restrict__ Cm, int64_t ldc,
    const int *__restrict__ row_idx, const int *__restrict__ n_rows,
    unsigned long long *__restrict__ flop_counter) {
  __shared__ __attribute__((aligned(16))) double lds[2][2][GEMM_BM][GEMM_LDS];  // [stage][A|B]
  __shared__ int s_rows[GEMM_BM];
  if (n_rows) M = *n_rows;
  if (flop_counter && blockIdx.x == 0 && threadIdx.x == 0)  // algorithmic flops of this launch
    atomicAdd(flop_counter, (unsigned long long)(2 * M * N * K));
  const int Tm = (int)((M + GEMM_BM - 1) / GEMM_BM), Tn = (int)((N + GEMM_BN - 1) / GEMM_BN);
  const GemmTileMap tile = gemm_tile_of_block(blockIdx.x, Tm, Tn);
  if (!tile.valid) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)tile.tm * GEMM_BM, n0 = (int64_t)tile.tn * GEMM_BN;
  if (tid < GEMM_BM) {
    const int64_t r = m0 + tid;
    s_rows[tid] = r < M ? (row_idx ? row_idx[r] : (int)r) : -1;
  }
  __syncthreads();
  const int64_t a_row = s_rows[tid >> 1];
  const int64_t b_row = (n0 + (tid >> 1)) < N ? n0 + (tid >> 1) : -1;

  d4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
      if (MODE == 1) {
        const int64_t col = n0 + (wave & 1) * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int64_t row = s_rows[(wave >> 1) * 64 + i * 16 + (lane >> 4) + 4 * r];
          if (row >= 0 && col < N) acc[i][j][r] = Cm[row * ldc + col];
        }
      }
    }

  double ra[8], rb[8];
  const int nk = (int)((K + GEMM_BK - 1) / GEMM_BK);
  const int srow = tid >> 1, scol = (tid & 1) * 8;

  gemm_load_tile<VEC>(A, lda, a_row, 0, K, tid, ra);
  gemm_load_tile<VEC>(B, ldb, b_row, 0, K, tid, rb);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    *reinterpret_cast<d2_t *>(&lds[0][0][srow][scol + 2 * i]) = (d2_t){ra[2 * i], ra[2 * i + 1]};
    *reinterpret_cast<d2_t *>(&lds[0][1][srow][scol + 2 * i]) = (d2_t){rb[2 * i], rb[2 * i + 1]};
  }
  __syncthreads();

  const int fr = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; kt++) {
    const int st = kt & 1;
    if (kt + 1 < nk) {
      gemm_load_tile<VEC>(A, lda, a_row, (int64_t)(kt + 1) * GEMM_BK, K, tid, ra);
      gemm_load_tile<VEC>(B, ldb, b_row, (int64_t)(kt + 1) * GEMM_BK, K, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; kk++) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = (MODE ? -1.0 : 1.0) * lds[st][0][wm * 64 + i * 16 + fr][kk * 4 + fk];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = lds[st][1][wn * 64 + j * 16 + fr][kk * 4 + fk];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        *reinterpret_cast<d2_t *>(&lds[st ^ 1][0][srow][scol + 2 * i]) =
            (d2_t){ra[2 * i], ra[2 * i + 1]};
        *reinterpret_cast<d2_t *>(&lds[st ^ 1][1][srow][scol + 2 * i]) =
            (d2_t){rb[2 * i], rb[2 * i + 1]};
      }
    }
    __syncthreads();
  }

  // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t col = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t row = s_rows[wm * 64 + i * 16 + fk + 4 * r];
        if (row >= 0 && col < N) Cm[row * ldc + col] = acc[i][j][r];
      }
    }
}

// ---- stream-K variant -------------------------------------------------------------
// A persistent grid of G workgroups (2 per CU).  A launch whose tile count is not a multiple
// of G (e.g. 2900 live chains: 1817 tiles = 3.55 rounds of 512) would pay for a partly empty
// last round; here all but the last 1..2 rounds run as whole tiles in the XCD-aware strided
// order and the remaining G..2G-1 tiles are cut into even contiguous (tile, k) ranges, so every
// workgroup finishes together.  A tile cut between two neighbouring workgroups is summed IN
// ORDER: the workgroup that owns the head of the tile (k = 0 .. ke) computes it first and
// publishes its accumulators; the owner of the tail loads them as its initial accumulator and
// continues the same k-chain -- bitwise the same result as the unsplit kernel, no atomics.
// Hand-off: plain stores -> s_waitcnt vmcnt(0) -> barrier -> agent-scope release -> flag;
// consumer: relaxed poll -> agent-scope acquire -> barrier -> plain loads
// (cdna_hip_programming.md, Guideline 16).  Every spin is bounded.  When fewer than G tiles
// exist (W < nk), or the last round is full to within 4 %, the grid strides over whole tiles.
struct GemmStreamK {
  double *partial;  // [G][16 * NJ][256] accumulators of head segments
  int *flags;       // [G] epoch of the last publish
  int *err;         // set to 1 if a bounded spin expired (host-visible)
  int epoch;
};

// PW: column tiles per panel.  The concurrent tiles of an XCD are a run of consecutive tile numbers, i.e. a
// (run / PW) x PW super-tile whose A and B panels its L2 shares.  128 x 128 tiles, 64 per XCD: 8 x 8.  128 x 256
// tiles, 32 per XCD (one workgroup per CU): PW = 4 gives 8 x 4 tiles = 1024 + 1024 operand rows per (XCD, round)
// where PW = 8 streamed 512 + 2048 (round 3; the order of the tiles does not change any result).
#ifndef AEHMC_GEMM_PW_WIDE
#define AEHMC_GEMM_PW_WIDE 4
#endif
template <int PW>
__device__ __forceinline__ void gemm_tile_coords(int t, int Tm, int Tn, int &tm, int &tn) {
  const int per_panel = Tm * PW;
  const int nfull = Tn / PW;
  const int panel = t / per_panel;
  if (panel < nfull) {
    const int rr = t - panel * per_panel;
    tn = panel * PW + rr % PW;
    tm = rr / PW;
  } else {
    int wlast = Tn - nfull * PW;
    if (wlast < 1) wlast = 1;
    const int rr = t - nfull * per_panel;
    tn = nfull * PW + rr % wlast;
    tm = rr / wlast;
  }
}

// NJ = 4: 128 x 128 tiles, two workgroups per CU.  NJ = 8: 128 x 256 tiles, one workgroup per CU
// whose four waves own 64 x 128 each (128 f64 accumulators per lane, in AGPRs): a third fewer
// operand bytes and LDS fragment reads per MFMA.
// PIPE: the software-pipelined K loop at one workgroup per CU (NJ = 8 always; NJ = 4 for launches
// with too few wide tiles to fill the CUs)
template <bool VEC, int NJ, bool PIPE = (NJ == 8)>
__global__ __launch_bounds__(256, (PIPE ? 1 : 2)) void gemm_nt_f64_streamk_kernel(
    int64_t M, int64_t N, int64_t K, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double *__restrict__ Cm, int64_t ldc,
    const int *__restrict__ row_idx, const int *__restrict__ n_rows,
    unsigned long long *__restrict__ flop_counter, GemmStreamK sk) {
  constexpr int BN = NJ * 32, NB = BN / 128;  // NB: 128-row groups of the B tile
  __shared__ __attribute__((aligned(16))) double ldsA[2][GEMM_BM][GEMM_LDS];
  __shared__ __attribute__((aligned(16))) double ldsB[2][BN][GEMM_LDS];
  __shared__ int s_rows[GEMM_BM];
  if (n_rows) M = *n_rows;
  if (flop_counter && blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(flop_counter, (unsigned long long)(2 * M * N * K));
  const int G = gridDim.x;                                     // multiple of 8
  const int bb = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;  // an XCD owns a contiguous tile range
  const int Tm = (int)((M + GEMM_BM - 1) / GEMM_BM), Tn = (int)((N + BN - 1) / BN);
  const long long T = (long long)Tm * Tn;
  if (T == 0) return;
  const int nk = (int)((K + GEMM_BK - 1) / GEMM_BK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int srow = tid >> 1, scol = (tid & 1) * 8;
  const int fr = lane & 15, fk = lane >> 4;

  // one pass over tile t, K-tiles [kb, ke): mode 0 = full tile -> C; 1 = head -> publish
  // partial; 2 = tail: continue from the neighbour's partial -> C
  auto pass = [&](int t, int kb, int ke, int mode) {
    int tm, tn;
    gemm_tile_coords<(NJ == 8 ? AEHMC_GEMM_PW_WIDE : GEMM_PANEL_W)>(t, Tm, Tn, tm, tn);
    const int64_t m0 = (int64_t)tm * GEMM_BM, n0 = (int64_t)tn * BN;
    __syncthreads();  // previous pass is done with s_rows / LDS
    if (tid < GEMM_BM) {
      const int64_t r = m0 + tid;
      s_rows[tid] = r < M ? (row_idx ? row_idx[r] : (int)r) : -1;
    }
    __syncthreads();
    const int64_t a_row = s_rows[tid >> 1];
    int64_t b_row[NB];
#pragma unroll
    for (int g = 0; g < NB; g++) b_row[g] = (n0 + g * 128 + (tid >> 1)) < N ? n0 + g * 128 + (tid >> 1) : -1;
    d4_t acc[4][NJ];
    if (mode == 2) {
      if (tid == 0) {
        const long long t0 = clock64();
        while (__hip_atomic_load(&sk.flags[bb - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
          __builtin_amdgcn_s_sleep(16);
          if (clock64() - t0 > 4000000000LL) {  // ~2 s: never hang the GPU
            *sk.err = 1;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const double *src = sk.partial + (size_t)(bb - 1) * (NJ * 16 * 256);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[i][j][r] = src[((i * NJ + j) * 4 + r) * 256 + tid];
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
    }
    if constexpr (PIPE) {
      // ---- software-pipelined K loop for the one-wave-per-SIMD tile (K % 16 == 0, 16-byte
      // aligned operands: checked by the launcher).  Nothing hides a stall here, so every
      // non-MFMA instruction is placed between MFMAs: fragments of k-step kk+1 are read while
      // the MFMAs of kk run, the next K-tile is fetched during kk = 0, stored to the other LDS
      // stage during kk = 2, and the barrier sits before kk = 3, whose MFMAs cover the first
      // fragment reads of the next tile.  Loads are branch-free: rows past M / N re-read a
      // valid row (their results are never stored), the prefetch index is clamped.
      const int64_t a_ld = a_row >= 0 ? a_row : (int64_t)s_rows[0];
      const double *pa = A + a_ld * lda + (tid & 1) * 8;
      const double *pb[NB];
#pragma unroll
      for (int g = 0; g < NB; g++) pb[g] = B + (b_row[g] >= 0 ? b_row[g] : N - 1) * ldb + (tid & 1) * 8;
      // Three register sets of global prefetch, used in rotation: a set is consumed (stored to LDS)
      // in phase 2 of every third tile and refilled, three loads per phase, over the four phases
      // that follow -- every load has at least two whole K-tiles (~8 us) to arrive, and a workgroup's demand
      // on L2 / HBM is even in time instead of a burst per tile (with bursts, the slowest of the
      // four waves stalls the barrier of every tile: 67 instead of 74 TFLOP/s).
      constexpr int NL = 4 + 4 * NB;  // 16-byte loads per thread and K-tile (12)
      d2_t gs[3][NL];
      auto gload1 = [&](auto set_tag, int l, int kt) {  // load number l of tile kt into set
        constexpr int S = decltype(set_tag)::value;
        const int ktc = kt < ke ? kt : ke - 1;
        if (l < 4) gs[S][l] = *reinterpret_cast<const d2_t *>(pa + (int64_t)ktc * GEMM_BK + 2 * l);
        else gs[S][l] = *reinterpret_cast<const d2_t *>(pb[(l - 4) / 4] + (int64_t)ktc * GEMM_BK + 2 * ((l - 4) % 4));
      };
      auto lstore = [&](auto set_tag, int st) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          *reinterpret_cast<d2_t *>(&ldsA[st][srow][scol + 2 * i]) = gs[S][i];
#pragma unroll
          for (int g = 0; g < NB; g++) *reinterpret_cast<d2_t *>(&ldsB[st][g * 128 + srow][scol + 2 * i]) = gs[S][4 + 4 * g + i];
        }
      };
      double fa[2][4], fb[2][NJ];
      auto fread = [&](int buf, int st, int kk) {
#pragma unroll
        for (int i = 0; i < 4; i++) fa[buf][i] = ldsA[st][wm * 64 + i * 16 + fr][kk * 4 + fk];
#pragma unroll
        for (int j = 0; j < NJ; j++) fb[buf][j] = ldsB[st][wn * (BN / 2) + j * 16 + fr][kk * 4 + fk];
      };
      auto mfmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
      };
      // scheduling recipe of one phase: n_a x {1 MFMA, 1 op of class a}, then n_b x {k MFMA, 1 op
      // of class b} spread over the remaining MFMAs
#define GEMM_SCHED(mask_a, n_a, mask_b, n_b)                                       \
  _Pragma("unroll") for (int z = 0; z < (n_a); z++) {                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             \
    __builtin_amdgcn_sched_group_barrier((mask_a), 1, 0);                          \
  }                                                                                \
  _Pragma("unroll") for (int z = 0; z < (n_b); z++) {                              \
    __builtin_amdgcn_sched_group_barrier(0x008, (4 * NJ - (n_a)) / ((n_b) + 1), 0); \
    __builtin_amdgcn_sched_group_barrier((mask_b), 1, 0);                          \
  }                                                                                \
  __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ - (n_a) - (n_b) * ((4 * NJ - (n_a)) / ((n_b) + 1)), 0);
      constexpr int Q = NL / 4;  // loads per phase
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      using S2 = std::integral_constant<int, 2>;
      // prologue: tile kb -> LDS stage 0 (through set 0); tile kb+1 -> set 0, tile kb+2 -> set 1,
      // first quarter of tile kb+3 -> set 2
#pragma unroll
      for (int l = 0; l < NL; l++) gload1(S0{}, l, kb);
      lstore(S0{}, 0);
#pragma unroll
      for (int l = 0; l < NL; l++) gload1(S0{}, l, kb + 1);
#pragma unroll
      for (int l = 0; l < NL; l++) gload1(S1{}, l, kb + 2);
#pragma unroll
      for (int l = 0; l < Q; l++) gload1(S2{}, l, kb + 3);
      __syncthreads();
      fread(0, 0, 0);
      // one K-tile: `cons` holds tile kt+1 and is stored to LDS in phase 2; `fill` (the set consumed
      // one tile earlier) takes the last three quarters of tile kt+3 in phases 0-2; after its store
      // `cons` takes the first quarter of tile kt+4
      auto tile = [&](auto cons, auto fill, int kt) {
        const int st = (kt - kb) & 1;
#pragma unroll
        for (int l = 0; l < Q; l++) gload1(fill, Q + l, kt + 3);  // phase 0
        fread(1, st, 1);
        mfmas(0);
        GEMM_SCHED(0x100, 4 + NJ, 0x020, Q)
#pragma unroll
        for (int l = 0; l < Q; l++) gload1(fill, 2 * Q + l, kt + 3);  // phase 1
        fread(0, st, 2);
        mfmas(1);
        GEMM_SCHED(0x100, 4 + NJ, 0x020, Q)
        lstore(cons, st ^ 1);  // phase 2
        fread(1, st, 3);
#pragma unroll
        for (int l = 0; l < Q; l++) gload1(fill, 3 * Q + l, kt + 3);
        mfmas(0);
        GEMM_SCHED(0x200, NL, 0x100, 4 + NJ)
        __syncthreads();
#pragma unroll
        for (int l = 0; l < Q; l++) gload1(cons, l, kt + 4);  // phase 3
        fread(0, st ^ 1, 0);
        mfmas(1);
        GEMM_SCHED(0x100, 4 + NJ, 0x020, Q)
      };
      for (int kt = kb; kt < ke; kt += 3) {
        tile(S0{}, S2{}, kt);
        if (kt + 1 < ke) tile(S1{}, S0{}, kt + 1);
        if (kt + 2 < ke) tile(S2{}, S1{}, kt + 2);
      }
#undef GEMM_SCHED
    } else {
    double ra[8], rb[NB][8];
      gemm_load_tile<VEC>(A, lda, a_row, (int64_t)kb * GEMM_BK, K, tid, ra);
  #pragma unroll
      for (int g = 0; g < NB; g++) gemm_load_tile<VEC>(B, ldb, b_row[g], (int64_t)kb * GEMM_BK, K, tid, rb[g]);
  #pragma unroll
      for (int i = 0; i < 4; i++) {
        *reinterpret_cast<d2_t *>(&ldsA[0][srow][scol + 2 * i]) = (d2_t){ra[2 * i], ra[2 * i + 1]};
  #pragma unroll
        for (int g = 0; g < NB; g++)
          *reinterpret_cast<d2_t *>(&ldsB[0][g * 128 + srow][scol + 2 * i]) = (d2_t){rb[g][2 * i], rb[g][2 * i + 1]};
      }
      __syncthreads();
      for (int kt = kb; kt < ke; kt++) {
        const int st = (kt - kb) & 1;
        if (kt + 1 < ke) {
          gemm_load_tile<VEC>(A, lda, a_row, (int64_t)(kt + 1) * GEMM_BK, K, tid, ra);
  #pragma unroll
          for (int g = 0; g < NB; g++)
            gemm_load_tile<VEC>(B, ldb, b_row[g], (int64_t)(kt + 1) * GEMM_BK, K, tid, rb[g]);
        }
  #pragma unroll
        for (int kk = 0; kk < GEMM_BK / 4; kk++) {
          double a[4], b[NJ];
  #pragma unroll
          for (int i = 0; i < 4; i++) a[i] = ldsA[st][wm * 64 + i * 16 + fr][kk * 4 + fk];
  #pragma unroll
          for (int j = 0; j < NJ; j++) b[j] = ldsB[st][wn * (BN / 2) + j * 16 + fr][kk * 4 + fk];
  #pragma unroll
          for (int i = 0; i < 4; i++)
  #pragma unroll
            for (int j = 0; j < NJ; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ke) {
  #pragma unroll
          for (int i = 0; i < 4; i++) {
            *reinterpret_cast<d2_t *>(&ldsA[st ^ 1][srow][scol + 2 * i]) = (d2_t){ra[2 * i], ra[2 * i + 1]};
  #pragma unroll
            for (int g = 0; g < NB; g++)
              *reinterpret_cast<d2_t *>(&ldsB[st ^ 1][g * 128 + srow][scol + 2 * i]) = (d2_t){rb[g][2 * i], rb[g][2 * i + 1]};
          }
        }
        __syncthreads();
      }
}
    if (mode == 1) {
      double *dst = sk.partial + (size_t)bb * (NJ * 16 * 256);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) dst[((i * NJ + j) * 4 + r) * 256 + tid] = acc[i][j][r];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&sk.flags[bb], sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int64_t col = n0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int64_t row = s_rows[wm * 64 + i * 16 + fk + 4 * r];
            if (row >= 0 && col < N) Cm[row * ldc + col] = acc[i][j][r];
          }
        }
    }
  };

  const long long total = T * nk;
  const long long W = (total + G - 1) / G;
  // Whole tiles, strided over the grid (the one-tile-per-workgroup order, which keeps the
  // workgroups of an XCD on one column panel at a time) when there are fewer tiles than
  // workgroups or when the last round is full to within 4 %; contiguous (tile, k) ranges cost
  // L2 locality (measured: -11 % at 4.94 rounds) and only pay when part of a round would be wasted.
  const long long rounds = (T + G - 1) / G;
  if (T < G || W < nk || (rounds * G - T) * 100 < 4 * (long long)G) {
    for (long long base = 0; base < T; base += G) {
      // same tile -> XCD assignment as gemm_tile_of_block within each round: XCD x (= workgroup
      // index mod 8) takes tiles [x chunk, (x+1) chunk) of the round.  Every workgroup index below
      // 8 chunk takes part -- NOT only those below `left`: with left % 8 != 0 the last slot of
      // the higher XCDs would never be visited and their tiles would keep stale output.
      const long long left = (T - base < G) ? T - base : G;
      const long long chunk = (left + 7) / 8, slot = blockIdx.x / 8, x = blockIdx.x % 8;
      if (slot < chunk && x * chunk + slot < left) pass((int)(base + x * chunk + slot), 0, nk, 0);
    }
    return;
  }
  // Hybrid schedule: all but the last 1..2 rounds run as whole tiles in the strided order above
  // (an XCD's workgroups share operand panels in L2); only the remaining G..2G-1 tiles are cut
  // into even contiguous (tile, k) ranges, so every workgroup ends at the same time and a tile
  // is shared by at most three workgroups.
  const long long Tdp = (T / G - 1) * G;  // >= 0: the launcher requires T >= G
  for (long long t = blockIdx.x; t < Tdp; t += G) {
    const long long base = (t / G) * G, chunk = G / 8, slot = (t - base) / 8, x = (t - base) % 8;
    pass((int)(base + x * chunk + slot), 0, nk, 0);
  }
  const long long total_sk = (T - Tdp) * nk, Wsk = (total_sk + G - 1) / G;
  const long long it0 = Tdp * nk + (long long)bb * Wsk;
  const long long it1 = (it0 + Wsk < total) ? it0 + Wsk : total;
  if (it0 >= it1) return;
  const int tf = (int)(it0 / nk), ks = (int)(it0 % nk);
  const int tl = (int)((it1 - 1) / nk), ke = (int)((it1 - 1) % nk) + 1;
  const bool has_tail = ks > 0;             // tile tf: [ks, nk) continues workgroup bb-1's head
  const bool has_head = ke < nk;            // tile tl: [0, ke) is continued by workgroup bb+1
  if (has_head) pass(tl, 0, ke, 1);         // first: the neighbour waits for it last
  for (int t = has_tail ? tf + 1 : tf; t <= (has_head ? tl - 1 : tl); t++) pass(t, 0, nk, 0);
  if (has_tail) pass(tf, ks, nk, 2);
}

// ---- tail kernel: a handful of rows (M <= 16 NI) --------------------------------------
// Late in a NUTS transition only the deepest trees are still alive; the product is then bound
// by streaming B (D x D) once, not by MFMAs, and one 128 x 128 tile per workgroup would put
// 79 workgroups on the GPU with one K-tile in flight each (1.07 ms for 8 rows at D = 1e4).
// Here one WAVE owns 16 output columns (16 rows of B) and all NI row blocks: per K-tile it
// fetches its 2 KB of B and NI x 2 KB of A in full 128-byte lines (lane = (row, 32-byte
// quarter)), P K-tiles ahead in registers, passes them through a wave-private LDS tile into
// MFMA fragment order and issues 4 NI MFMAs -- the same MFMA sequence per 16 x 16 block as
// every other kernel here, hence the same bits.  No barriers; ~2.4 waves per CU.
template <int NI, int P>
__global__ __launch_bounds__(256) void gemm_nt_f64_tail_kernel(
    int64_t M, int64_t N, int64_t K, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double *__restrict__ Cm, int64_t ldc,
    const int *__restrict__ row_idx, const int *__restrict__ n_rows,
    unsigned long long *__restrict__ flop_counter) {
  __shared__ __attribute__((aligned(16))) double tB[4][16][GEMM_LDS];
  __shared__ __attribute__((aligned(16))) double tA[4][NI][16][GEMM_LDS];
  if (n_rows) M = *n_rows;
  if (flop_counter && blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(flop_counter, (unsigned long long)(2 * M * N * K));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (M <= 0 || n0 >= N) return;
  const int r = lane >> 2, kq = lane & 3;  // staging role: row r, doubles 4 kq .. 4 kq + 3 of the K-tile
  const int fr = lane & 15, fk = lane >> 4;  // MFMA role
  const double *pb = B + ((n0 + r) < N ? n0 + r : N - 1) * ldb + 4 * kq;
  const double *pa[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int64_t m = 16 * i + r;
    const int64_t row = m < M ? (row_idx ? (int64_t)row_idx[m] : m) : (row_idx ? (int64_t)row_idx[0] : 0);
    pa[i] = A + row * lda + 4 * kq;
  }
  const int nk = (int)((K + GEMM_BK - 1) / GEMM_BK);
  d2_t gb[P][2], ga[P][NI][2];
  auto fetch = [&](int s, int kt) {  // K-tile kt -> register stage s (zeros past K; kt clamped past the end)
    const int ktc = kt < nk ? kt : nk - 1;
    const int64_t k0 = (int64_t)ktc * GEMM_BK + 4 * kq;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const bool inb = k0 + 2 * h + 2 <= K;
      const int64_t off = (int64_t)ktc * GEMM_BK + (inb ? 2 * h : -4 * kq);  // in bounds: column 0 of the tile
      d2_t v = *reinterpret_cast<const d2_t *>(pb + off);
      gb[s][h] = inb ? v : (d2_t){(k0 + 2 * h < K) ? pb[(int64_t)ktc * GEMM_BK + 2 * h] : 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < NI; i++) {
        d2_t w = *reinterpret_cast<const d2_t *>(pa[i] + off);
        ga[s][i][h] = inb ? w : (d2_t){(k0 + 2 * h < K) ? pa[i][(int64_t)ktc * GEMM_BK + 2 * h] : 0.0, 0.0};
      }
    }
  };
  d4_t acc[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < P; s++) fetch(s, s);
  for (int kt0 = 0; kt0 < nk; kt0 += P) {
#pragma unroll
    for (int s = 0; s < P; s++) {
      const int kt = kt0 + s;
      if (kt < nk) {  // wave-uniform
#pragma unroll
        for (int h = 0; h < 2; h++) {
          *reinterpret_cast<d2_t *>(&tB[wave][r][4 * kq + 2 * h]) = gb[s][h];
#pragma unroll
          for (int i = 0; i < NI; i++) *reinterpret_cast<d2_t *>(&tA[wave][i][r][4 * kq + 2 * h]) = ga[s][i][h];
        }
        fetch(s, kt + P);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS is in order within a wave
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 4; kk++) {
          const double b = tB[wave][fr][kk * 4 + fk];
#pragma unroll
          for (int i = 0; i < NI; i++)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(tA[wave][i][fr][kk * 4 + fk], b, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
  const int64_t col = n0 + fr;
#pragma unroll
  for (int i = 0; i < NI; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int64_t m = 16 * i + fk + 4 * q;
      if (m < M && col < N) Cm[(row_idx ? (int64_t)row_idx[m] : m) * ldc + col] = acc[i][q];
    }
}

// ---- small tiles for mid-size problems --------------------------------------------------------
// Chain-batched products of a mid-size problem (D = 65 .. ~1000: N = K = D, M = a few thousand chains) are only a
// few dozen 128 x 128 tiles -- D = 200 with 4096 chains: 64 workgroups on 256 CUs, 32 us per product.  Here a
// workgroup (4 waves, 2 x 2) computes a (32 I) x (32 J) tile, each wave I x J MFMA tiles: 32 x 64 or 64 x 64 tiles
// give every CU two or more workgroups.  An fp64 MFMA (16x16x4: 32 cycles) is slow against the LDS reads that feed it,
// so the small tile's lower operand reuse costs little; the operands are L2-resident at this size.  Same K-tile
// (16), same per-element summation order as every other variant (bitwise the same result), rows compacted the same
// way; K may end inside a K-tile (zeros).
template <int I, int J, bool VEC, int KT = 2>
__global__ __launch_bounds__(256) void gemm_nt_f64_small_kernel(
    int64_t M, int64_t N, int64_t K, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double *__restrict__ Cm, int64_t ldc,
    const int *__restrict__ row_idx, const int *__restrict__ n_rows,
    unsigned long long *__restrict__ flop_counter) {
  constexpr int BM = 32 * I, BN = 32 * J;
  constexpr int LW = 16 * KT + 2;  // KT K-tiles of 16 per LDS stage and barrier
  __shared__ __attribute__((aligned(16))) double la[2][BM][LW];
  __shared__ __attribute__((aligned(16))) double lb[2][BN][LW];
  __shared__ int s_rows[BM];
  if (n_rows) M = *n_rows;
  if (flop_counter && blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(flop_counter, (unsigned long long)(2 * M * N * K));
  // block ids are dealt to the 8 XCDs round-robin: every XCD gets a contiguous run of tiles (row-major, the Tn
  // column tiles of a row block together), so a row block of A is fetched into ONE L2 instead of up to 8
  const int Tn = (int)((N + BN - 1) / BN), Tm = (int)((M + BM - 1) / BM);
  const int chunk = (Tm * Tn + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * chunk + (int)(blockIdx.x / 8);
  if ((int)(blockIdx.x / 8) >= chunk || tile >= Tm * Tn) return;
  const int64_t m0 = (int64_t)(tile / Tn) * BM, n0 = (int64_t)(tile % Tn) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  if (tid < BM) {
    const int64_t r = m0 + tid;
    s_rows[tid] = r < M ? (row_idx ? row_idx[r] : (int)r) : -1;
  }
  __syncthreads();
  // staging role: rows tid / 8 + 32 l, doubles 2 (tid % 8), +1 of the K-tile
  const int srow = tid >> 3, scol = (tid & 7) * 2;
  // (rows beyond M / N read a valid row instead: their products land in output elements that are never stored)
  const double *pa[I], *pb[J];
#pragma unroll
  for (int l = 0; l < I; l++) {
    const int r = s_rows[srow + 32 * l];
    pa[l] = A + (int64_t)(r >= 0 ? r : s_rows[0]) * lda + scol;
  }
#pragma unroll
  for (int l = 0; l < J; l++) {
    const int64_t r = n0 + srow + 32 * l;
    pb[l] = B + (r < N ? r : n0) * ldb + scol;
  }
  // branch-free: every load is issued (address clamped into the row), elements at k >= K are zeroed afterwards
  auto fetch = [&](const double *ptr, int64_t k0) -> d2_t {
    const int64_t k = k0 + scol;
    if (VEC) {  // K even: a pair is inside or outside as a whole
      const d2_t v = *reinterpret_cast<const d2_t *>(ptr + (k < K ? k0 : -(int64_t)scol));
      return k < K ? v : (d2_t){0.0, 0.0};
    }
    const double x0 = ptr[k < K ? k0 : -(int64_t)scol], x1 = ptr[k + 1 < K ? k0 + 1 : -(int64_t)scol];
    return (d2_t){k < K ? x0 : 0.0, k + 1 < K ? x1 : 0.0};
  };
  d4_t acc[I][J];
#pragma unroll
  for (int i = 0; i < I; i++)
#pragma unroll
    for (int j = 0; j < J; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  // operands are fetched P stages of KT K-tiles ahead into a ring of register stages.  (Measured, rocprofv3: K-tiles of
  // 16 or 32 per barrier, 2 or 4 ring stages, loads kept out of branches so that the waits before the LDS writes are
  // partial instead of s_waitcnt vmcnt(0) -- all within 5 %: at 28 % (D = 200) to 52 % (D = 500) of the 78.6 TFLOP/s
  // fp64 MFMA peak the rest is the fixed cost of a launch this short, not the K loop.)
  constexpr int P = 4 / KT < 2 ? 2 : 4 / KT, BKS = GEMM_BK * KT;  // ring stages; K per stage
  d2_t ra[P][KT][I], rb[P][KT][J];
  const int nk = (int)((K + BKS - 1) / BKS);
#pragma unroll
  for (int s = 0; s < P; s++)
#pragma unroll
    for (int u = 0; u < KT; u++) {
#pragma unroll
      for (int l = 0; l < I; l++) ra[s][u][l] = fetch(pa[l], (int64_t)s * BKS + u * GEMM_BK);
#pragma unroll
      for (int l = 0; l < J; l++) rb[s][u][l] = fetch(pb[l], (int64_t)s * BKS + u * GEMM_BK);
    }
  const int fr = lane & 15, fk = lane >> 4;
  for (int kt0 = 0; kt0 < nk; kt0 += P) {
#pragma unroll
    for (int s = 0; s < P; s++) {
      const int kt = kt0 + s;
      if (kt < nk) {  // workgroup-uniform
        const int st = s & 1;  // (kt0 is a multiple of the even P)
        // LDS stage st was last read in iteration kt - 2; every wave has passed iteration kt - 1's barrier since
#pragma unroll
        for (int u = 0; u < KT; u++) {
#pragma unroll
          for (int l = 0; l < I; l++)
            *reinterpret_cast<d2_t *>(&la[st][srow + 32 * l][u * GEMM_BK + scol]) = ra[s][u][l];
#pragma unroll
          for (int l = 0; l < J; l++)
            *reinterpret_cast<d2_t *>(&lb[st][srow + 32 * l][u * GEMM_BK + scol]) = rb[s][u][l];
        }
#pragma unroll
        for (int u = 0; u < KT; u++) {
          const int64_t k0 = (int64_t)(kt + P) * BKS + u * GEMM_BK;  // (past K: zeros)
#pragma unroll
          for (int l = 0; l < I; l++) ra[s][u][l] = fetch(pa[l], k0);
#pragma unroll
          for (int l = 0; l < J; l++) rb[s][u][l] = fetch(pb[l], k0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BKS / 4; kk++) {
          double a[I], b[J];
#pragma unroll
          for (int i = 0; i < I; i++) a[i] = la[st][wm * 16 * I + i * 16 + fr][kk * 4 + fk];
#pragma unroll
          for (int j = 0; j < J; j++) b[j] = lb[st][wn * 16 * J + j * 16 + fr][kk * 4 + fk];
#pragma unroll
          for (int i = 0; i < I; i++)
#pragma unroll
            for (int j = 0; j < J; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < I; i++)
#pragma unroll
    for (int j = 0; j < J; j++) {
      const int64_t col = n0 + wn * 16 * J + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = s_rows[wm * 16 * I + i * 16 + fk + 4 * r];
        if (row >= 0 && col < N) Cm[(int64_t)row * ldc + col] = acc[i][j][r];
      }
    }
}

inline hipError_t launch_gemm_nt_f64(int64_t M, int64_t N, int64_t K, const double *A,
                                     int64_t lda, const double *B, int64_t ldb, double *Cm,
                                     int64_t ldc, hipStream_t stream,
                                     const int *row_idx = nullptr, const int *n_rows = nullptr,
                                     unsigned long long *flop_counter = nullptr,
                                     const GemmStreamK *sk = nullptr, int sk_grid = 0, int mode = 0,
                                     int sk_grid_wide = 0, int small_tiles = 1) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int Tm = (int)((M + GEMM_BM - 1) / GEMM_BM), Tn = (int)((N + GEMM_BN - 1) / GEMM_BN);
  const int total = Tm * Tn;
  const int grid = ((total + 7) / 8) * 8;
  const bool vec = (lda % 2 == 0) && (ldb % 2 == 0) && ((uintptr_t)A % 16 == 0) &&
                   ((uintptr_t)B % 16 == 0);
  if (mode == 0 && vec && M <= 128) {  // a few rows: bandwidth-bound tail kernel, one wave per 16 columns
    const dim3 tg((unsigned)((N + 63) / 64));
    if (M <= 16)
      hipLaunchKernelGGL((gemm_nt_f64_tail_kernel<1, 6>), tg, dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc,
                         row_idx, n_rows, flop_counter);
    else if (M <= 32)
      hipLaunchKernelGGL((gemm_nt_f64_tail_kernel<2, 4>), tg, dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc,
                         row_idx, n_rows, flop_counter);
    else if (M <= 64)
      hipLaunchKernelGGL((gemm_nt_f64_tail_kernel<4, 3>), tg, dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc,
                         row_idx, n_rows, flop_counter);
    else
      hipLaunchKernelGGL((gemm_nt_f64_tail_kernel<8, 2>), tg, dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc,
                         row_idx, n_rows, flop_counter);
    return hipGetLastError();
  }
  if (mode == 0 && small_tiles && M > 128 && N <= 2048 && K >= 2 && total < 256) {
    // mid-size problem, the 128 x 128 tiles would leave CUs idle: the largest small tile that still gives every CU two
    // workgroups, else the smallest
    const auto count = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
#define AEHMC_GEMM_SMALL(II, JJ)                                                                                  \
  do {                                                                                                            \
    if (vec && K % 2 == 0)                                                                                        \
      hipLaunchKernelGGL((gemm_nt_f64_small_kernel<II, JJ, true>), dim3((unsigned)((count(32 * II, 32 * JJ) + 7) / 8 * 8)),        \
                         dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);  \
    else                                                                                                          \
      hipLaunchKernelGGL((gemm_nt_f64_small_kernel<II, JJ, false>), dim3((unsigned)((count(32 * II, 32 * JJ) + 7) / 8 * 8)),       \
                         dim3(256), 0, stream, M, N, K, A, lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);  \
    return hipGetLastError();                                                                                     \
  } while (0)
    if (small_tiles == 3 || (small_tiles == 1 && count(64, 128) >= 512)) AEHMC_GEMM_SMALL(2, 4);
    if (small_tiles == 2 || (small_tiles == 1 && count(64, 64) >= 512)) AEHMC_GEMM_SMALL(2, 2);
    AEHMC_GEMM_SMALL(1, 2);
#undef AEHMC_GEMM_SMALL
  }
  if (mode == 1) {
    hipLaunchKernelGGL((gemm_nt_f64_kernel<false, 1>), dim3(grid), dim3(256), 0, stream, M, N, K, A, lda,
                       B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);
    return hipGetLastError();
  }
  if (mode == 2) {
    hipLaunchKernelGGL((gemm_nt_f64_kernel<false, 2>), dim3(grid), dim3(256), 0, stream, M, N, K, A, lda,
                       B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);
    return hipGetLastError();
  }
  if (vec && sk && sk_grid_wide > 0 && K % GEMM_BK == 0 && Tm * (int)((N + 255) / 256) >= sk_grid_wide) {  // 128 x 256 tiles
    hipLaunchKernelGGL((gemm_nt_f64_streamk_kernel<true, 8>), dim3(sk_grid_wide), dim3(256), 0, stream, M, N, K,
                       A, lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter, *sk);
    return hipGetLastError();
  }
  if (vec && sk && sk_grid_wide > 0 && K % GEMM_BK == 0 && M > 128) {
    // 129 .. ~768 rows: too few 128 x 256 tiles for the CUs -- 128 x 128 tiles on the pipelined loop, one
    // workgroup per CU (one tile each, or even (tile, k) ranges once there are more tiles than CUs)
    hipLaunchKernelGGL((gemm_nt_f64_streamk_kernel<true, 4, true>), dim3(sk_grid_wide), dim3(256), 0, stream, M, N,
                       K, A, lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter, *sk);
    return hipGetLastError();
  }
  if (vec && sk && sk_grid > 0 && total >= sk_grid) {  // enough tiles for an even (tile, k) split
    hipLaunchKernelGGL((gemm_nt_f64_streamk_kernel<true, 4>), dim3(sk_grid), dim3(256), 0, stream, M, N, K,
                       A, lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter, *sk);
    return hipGetLastError();
  }
  if (vec)
    hipLaunchKernelGGL(gemm_nt_f64_kernel<true>, dim3(grid), dim3(256), 0, stream, M, N, K, A,
                       lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);
  else
    hipLaunchKernelGGL(gemm_nt_f64_kernel<false>, dim3(grid), dim3(256), 0, stream, M, N, K, A,
                       lda, B, ldb, Cm, ldc, row_idx, n_rows, flop_counter);
  return hipGetLastError();
}

}  // namespace aehmc
