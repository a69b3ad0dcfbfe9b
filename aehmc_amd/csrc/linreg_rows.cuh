// Row sums of the regression target (examples/LinearRegression.ipynb:126-166, q = [w, log n]) for
// the workgroup-cooperative kernels: a 512-thread workgroup evaluates sum(x r) and sum(r^2),
// r = y - x w_k, over all N data rows for the FOUR chains k it owns; every thread accumulates its
// share of the rows, the callers finish with wave sums and a fixed-order pass over the waves.
#pragma once
#include <hip/hip_runtime.h>

namespace aehmc {

constexpr int LR_BLOCK = 512, LR_WAVES = LR_BLOCK / 64;  // 4 chain waves + 4 waves that only serve rows
constexpr int LR_CHUNK = 256, LR_RING = 4;               // rows per chunk, chunks in flight per wave (128 KB of LDS)
constexpr size_t LR_RING_BYTES = (size_t)LR_WAVES * LR_RING * LR_CHUNK * 2 * sizeof(double);

// Rows streamed from L2 (any N): `dyn_lds` holds LR_RING_BYTES.
__device__ __forceinline__ void lr_rows_stream(const double *X, const double *y, long long N, double *dyn_lds,
                                         int wave, int lane, const double (&w4)[4], double (&sxr)[4],
                                         double (&srr)[4]) {
  // The rows stream through a wave-private LDS ring filled by LDS-DMA (global_load_lds, 16 B per
  // lane: one instruction lands 128 consecutive doubles): a chunk is 256 rows of X and of y
  // (4 x 1 KB), LR_RING chunks per wave are in flight, so a chunk has ~3 chunk-times (> 1 us) to
  // arrive and no VGPR holds data in flight.  Lane l adds rows r0+l, r0+64+l, r0+128+l, r0+192+l of
  // its wave's chunks in ascending order.
  {
    double *const ring = dyn_lds + (size_t)wave * (LR_RING * LR_CHUNK * 2);
    const int nchunks = (int)(N / LR_CHUNK);                      // full chunks
    const int nm = wave < nchunks ? (nchunks - wave + LR_WAVES - 1) / LR_WAVES : 0;  // this wave's
    auto issue = [&](int m) {
      const long long r0 = (long long)(wave + LR_WAVES * m) * LR_CHUNK;
      double *slot = ring + (size_t)(m % LR_RING) * (LR_CHUNK * 2);
  #pragma unroll
      for (int h = 0; h < LR_CHUNK / 128; h++) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(X + r0 + 128 * h + 2 * lane),
            (__attribute__((address_space(3))) void *)(slot + 128 * h), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(y + r0 + 128 * h + 2 * lane),
            (__attribute__((address_space(3))) void *)(slot + LR_CHUNK + 128 * h), 16, 0, 0);
      }
    };
    constexpr int PER = 2 * (LR_CHUNK / 128);  // DMA instructions per chunk
    for (int m = 0; m < LR_RING - 1 && m < nm; m++) issue(m);
    for (int m = 0; m < nm; m++) {
      if (m + LR_RING - 1 < nm) {
        issue(m + LR_RING - 1);  // into the slot read in the previous step (its values are in registers)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (((LR_RING - 1) * PER) & 0xF) | ((((LR_RING - 1) * PER) >> 4) << 14));
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tail of the stream
      }
      __builtin_amdgcn_sched_barrier(0);
      const double *slot = ring + (size_t)(m % LR_RING) * (LR_CHUNK * 2);
      double xs[LR_CHUNK / 64], ys[LR_CHUNK / 64];
  #pragma unroll
      for (int u = 0; u < LR_CHUNK / 64; u++) {
        xs[u] = slot[64 * u + lane];
        ys[u] = slot[LR_CHUNK + 64 * u + lane];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the slot may be refilled from here on
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int u = 0; u < LR_CHUNK / 64; u++)
  #pragma unroll
        for (int k = 0; k < 4; k++) {
          const double rr = ys[u] - xs[u] * w4[k];
          sxr[k] += xs[u] * rr;
          srr[k] += rr * rr;
        }
    }
    for (long long i = (long long)nchunks * LR_CHUNK + threadIdx.x; i < N; i += LR_BLOCK) {  // last < 256 rows
      const double x = X[i], yy = y[i];
  #pragma unroll
      for (int k = 0; k < 4; k++) {
        const double rr = yy - x * w4[k];
        sxr[k] += x * rr;
        srr[k] += rr * rr;
      }
    }
  }
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
    for (int k = 0; k < K; k++) {
      const double rr = yy - x * w4[k];
      sxr[k] += x * rr;
      srr[k] += rr * rr;
    }
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
