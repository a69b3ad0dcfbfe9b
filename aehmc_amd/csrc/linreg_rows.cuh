// Row sums of the regression target (examples/LinearRegression.ipynb:126-166, q = [w, log n]) for
// the workgroup-cooperative kernels: a 512-thread workgroup evaluates sum(x r) and sum(r^2),
// r = y - x w_k, over all N data rows for the FOUR chains k it owns; every thread accumulates its
// share of the rows, the callers finish with wave sums and a fixed-order pass over the waves.
#pragma once
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime, the math functions and the fixed-width integers itself) */
#include <hip/hip_runtime.h>
#endif

namespace aehmc {

constexpr int LR_BLOCK = 512, LR_WAVES = LR_BLOCK / 64;  // 4 chain waves + 4 waves that only serve rows

// One row's terms for one chain: r = y - x w, sxr += x r, srr += r r as three fused multiply-adds
// (one rounding per term where separate multiplies and adds have two; every regression kernel
// uses this same form, so their sums differ only by the order of the rows).
__device__ __forceinline__ void lr_term(double x, double yy, double w, double &sxr, double &srr) {
  const double rr = __builtin_fma(-x, w, yy);
  sxr = __builtin_fma(x, rr, sxr);
  srr = __builtin_fma(rr, rr, srr);
}

// Rows streamed from L2 straight into registers (any N; X and y 16-byte aligned): a wavefront
// takes blocks of 64 x LR_UN 16-byte pieces (two rows each) of X and of y, blocks w, w + 8, ... in
// ascending order, the next block's loads in flight while the current one is used (two register
// buffers).  Measured on MI355X (tools/debug/lr_stream_bench.hip, 256 workgroups x 1e5 rows x 4
// chains): 16 us per sweep against 32 us for round 1's LDS-DMA ring with unfused arithmetic (23 us for these loads unfused); the
// loads alone take 12 us (every CU pulls all rows through its 64 B/clk vector-memory path).
// Adds this thread's rows to sxr[k] (sum(x r) of chain k) and srr[k] (sum(r^2)).
constexpr int LR_UN = 4;
// `between()` runs after the first block's loads have been issued (work that hides their latency).
template <class Between>
__device__ __forceinline__ void lr_rows_direct(const double *X, const double *y, long long N, int wave, int lane,
                                               const double (&w4)[4], double (&sxr)[4], double (&srr)[4],
                                               Between between) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *X2 = reinterpret_cast<const d2 *>(X), *y2 = reinterpret_cast<const d2 *>(y);
  constexpr int BLK = 64 * LR_UN;  // pieces per block
  const int nblk = (int)((N / 2) / BLK);
  d2 xa[LR_UN], ya[LR_UN], xb[LR_UN], yb[LR_UN];
  auto load = [&](int b, d2(&xx)[LR_UN], d2(&yy)[LR_UN]) {
    const long long p0 = (long long)b * BLK + lane;
#pragma unroll
    for (int u = 0; u < LR_UN; u++) {
      xx[u] = X2[p0 + 64 * u];
      yy[u] = y2[p0 + 64 * u];
    }
  };
  auto use = [&](const d2(&xx)[LR_UN], const d2(&yy)[LR_UN]) {
#pragma unroll
    for (int u = 0; u < LR_UN; u++)
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 4; k++) lr_term(xx[u][h], yy[u][h], w4[k], sxr[k], srr[k]);
  };
  int b = wave;
  if (b < nblk) load(b, xa, ya);
  between();
  for (; b < nblk; b += 2 * LR_WAVES) {
    const bool more = b + LR_WAVES < nblk;
    if (more) load(b + LR_WAVES, xb, yb);
    use(xa, ya);
    if (more) {
      if (b + 2 * LR_WAVES < nblk) load(b + 2 * LR_WAVES, xa, ya);
      use(xb, yb);
    }
  }
  for (long long i = (long long)nblk * BLK * 2 + wave * 64 + lane; i < N; i += LR_BLOCK) {  // last rows
    const double x = X[i], yy = y[i];
#pragma unroll
    for (int k = 0; k < 4; k++) lr_term(x, yy, w4[k], sxr[k], srr[k]);
  }
}
__device__ __forceinline__ void lr_rows_direct(const double *X, const double *y, long long N, int wave, int lane,
                                               const double (&w4)[4], double (&sxr)[4], double (&srr)[4]) {
  lr_rows_direct(X, y, N, wave, lane, w4, sxr, srr, [] {});
}

// Rows resident in LDS for the whole kernel (N * 16 bytes fit): lx = dyn_lds, ly = dyn_lds + N.
// Thread `tid` of `NT` adds rows tid, tid + NT, ... in ascending order for K <= 4 chains (compile
// time: a workgroup with a single chain -- the notebook's own run -- does a quarter of the
// arithmetic, and its chain wave takes no rows: NT = 448).
template <int K, int NT>
__device__ __forceinline__ void lr_rows_lds(const double *lx, const double *ly, long long N, int tid,
                                            const double (&w4)[4], double (&sxr)[4], double (&srr)[4]) {
  constexpr int UN = 4;  // rows per thread in flight
  constexpr int LR_BLOCK = NT;  // (shadows the workgroup size: the stride of this sweep)
  const long long nblk = N / (UN * LR_BLOCK);
  auto add = [&](double x, double yy) {
#pragma unroll
    for (int k = 0; k < K; k++) lr_term(x, yy, w4[k], sxr[k], srr[k]);
  };
  for (long long blk = 0; blk < nblk; blk++) {  // whole blocks: no bounds checks
    const long long i0 = blk * (UN * LR_BLOCK) + tid;
    double xs[UN], ys[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      xs[u] = lx[i0 + u * LR_BLOCK];
      ys[u] = ly[i0 + u * LR_BLOCK];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) add(xs[u], ys[u]);
  }
  for (long long i = nblk * (UN * LR_BLOCK) + tid; i < N; i += LR_BLOCK) add(lx[i], ly[i]);
}

// Wave totals of 8 per-lane values at once: three "transpose" stages (lane bit b keeps one half
// of the values and receives the partner's other half) leave ONE register per lane, three more
// butterfly stages finish it -- 10 cross-lane additions instead of the 48 of eight separate
// wave sums.  Afterwards lane l holds the wave total of v[l & 7].  Fixed order: deterministic.
__device__ __forceinline__ double dpp_xor1(double x) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ double dpp_xor2(double x) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x4E, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x4E, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_sum8(const double (&v)[8], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  double r2[4], r4[2];
#pragma unroll
  for (int i = 0; i < 4; i++) r2[i] = (b0 ? v[2 * i + 1] : v[2 * i]) + dpp_xor1(b0 ? v[2 * i] : v[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2; i++) r4[i] = (b1 ? r2[2 * i + 1] : r2[2 * i]) + dpp_xor2(b1 ? r2[2 * i] : r2[2 * i + 1]);
  double x = (b2 ? r4[1] : r4[0]) + __shfl_xor(b2 ? r4[0] : r4[1], 4);
  x += __shfl_xor(x, 8);
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

}  // namespace aehmc
