// Microbenchmark for the regression row sweep (c5): one workgroup evaluates sum(x r), sum(r^2),
// r = y - x w_k over N rows for its K chains, once per "sweep"; every workgroup streams the
// same (X, y) from L2.  Variants: LDS-DMA ring per wave (the product's linreg_rows.cuh form) and
// direct global loads into registers; K chains per workgroup; grid size; compute on/off.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/bin/lr_stream_bench tools/debug/lr_stream_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

template <int K, int WAVES, int RING, int CHUNK, bool COMPUTE>
__global__ __launch_bounds__(WAVES * 64) void k_dma(const double *X, const double *y, int N, int iters, double *out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *const ring = lds + (size_t)wave * (RING * CHUNK * 2);
  const int nchunks = N / CHUNK;
  const int nm = wave < nchunks ? (nchunks - wave + WAVES - 1) / WAVES : 0;
  double tot = 0.0;
  for (int it = 0; it < iters; it++) {
    double w[K], sxr[K], srr[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      w[k] = 3.0 + 1e-3 * k + 1e-6 * it + 1e-9 * blockIdx.x;
      sxr[k] = srr[k] = 0.0;
    }
    auto issue = [&](int m) {
      const long long r0 = (long long)(wave + WAVES * m) * CHUNK;
      double *slot = ring + (size_t)(m % RING) * (CHUNK * 2);
#pragma unroll
      for (int h = 0; h < CHUNK / 128; h++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + r0 + 128 * h + 2 * lane),
                                         (__attribute__((address_space(3))) void *)(slot + 128 * h), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(y + r0 + 128 * h + 2 * lane),
                                         (__attribute__((address_space(3))) void *)(slot + CHUNK + 128 * h), 16, 0, 0);
      }
    };
    constexpr int PER = 2 * (CHUNK / 128);
    static_assert((RING - 1) * PER < 64, "vmcnt range");
    for (int m = 0; m < RING - 1 && m < nm; m++) issue(m);
    for (int m = 0; m < nm; m++) {
      if (m + RING - 1 < nm) {
        issue(m + RING - 1);
        __builtin_amdgcn_s_waitcnt(0x0F70 | (((RING - 1) * PER) & 0xF) | ((((RING - 1) * PER) >> 4) << 14));
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      __builtin_amdgcn_sched_barrier(0);
      const double *slot = ring + (size_t)(m % RING) * (CHUNK * 2);
      double xs[CHUNK / 64], ys[CHUNK / 64];
#pragma unroll
      for (int u = 0; u < CHUNK / 64; u++) {
        xs[u] = slot[64 * u + lane];
        ys[u] = slot[CHUNK + 64 * u + lane];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      if (COMPUTE) {
#pragma unroll
        for (int u = 0; u < CHUNK / 64; u++)
#pragma unroll
          for (int k = 0; k < K; k++) {
            const double rr = ys[u] - xs[u] * w[k];
            sxr[k] += xs[u] * rr;
            srr[k] += rr * rr;
          }
      } else {
#pragma unroll
        for (int u = 0; u < CHUNK / 64; u++) sxr[0] += xs[u] + ys[u];
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) tot += sxr[k] + srr[k];
    __syncthreads();
  }
  if (tot == 12345.678) out[0] = tot;
}

// direct loads into registers: lane l of wave w takes 16-byte pieces (2 rows) of X and y
template <int K, int WAVES, int UN, bool COMPUTE, bool FMA = false, bool NT = true, int CACHED = 0>
__global__ __launch_bounds__(WAVES * 64) void k_reg(const double *X, const double *y, int N, int iters, double *out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *X2 = reinterpret_cast<const d2 *>(X), *y2 = reinterpret_cast<const d2 *>(y);
  const int npairs = N / 2;                 // 16-byte pieces
  constexpr int BLK = 64 * UN;              // pieces per wave-block
  const int nblk_all = npairs / BLK;
  const int nblk = nblk_all * (100 - CACHED) / 100;
  double tot = 0.0;
  double cx = 0.25 + 1e-3 * threadIdx.x, cy = 0.75;
  for (int it = 0; it < iters; it++) {
    double w[K], sxr[K], srr[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      w[k] = 3.0 + 1e-3 * k + 1e-6 * it + 1e-9 * blockIdx.x;
      sxr[k] = srr[k] = 0.0;
    }
    if (CACHED) {  // the cached rows: arithmetic on register data
      const int crow = (nblk_all - nblk) * BLK * 2 / (WAVES * 64);
      for (int r = 0; r < crow; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
          for (int k = 0; k < K; k++) {
            const double rr = __builtin_fma(-cx, w[k], cy);
            sxr[k] = __builtin_fma(cx, rr, sxr[k]);
            srr[k] = __builtin_fma(rr, rr, srr[k]);
          }
          asm volatile("" : "+v"(cx), "+v"(cy));
        }
      }
    }
    d2 xa[UN], ya[UN], xb[UN], yb[UN];
    auto load = [&](int b, d2 (&xx)[UN], d2 (&yy)[UN]) {
      const int p0 = b * BLK + lane;
#pragma unroll
      for (int u = 0; u < UN; u++) {
        xx[u] = NT ? __builtin_nontemporal_load(&X2[p0 + 64 * u]) : X2[p0 + 64 * u];
        yy[u] = NT ? __builtin_nontemporal_load(&y2[p0 + 64 * u]) : y2[p0 + 64 * u];
      }
    };
    auto use = [&](d2 (&xx)[UN], d2 (&yy)[UN]) {
      if (COMPUTE) {
#pragma unroll
        for (int u = 0; u < UN; u++)
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k = 0; k < K; k++) {
              const double x = xx[u][h], yv = yy[u][h];
              if (FMA) {
                const double rr = __builtin_fma(-x, w[k], yv);
                sxr[k] = __builtin_fma(x, rr, sxr[k]);
                srr[k] = __builtin_fma(rr, rr, srr[k]);
              } else {
                const double rr = yv - x * w[k];
                sxr[k] += x * rr;
                srr[k] += rr * rr;
              }
            }
      } else {
#pragma unroll
        for (int u = 0; u < UN; u++) sxr[0] += xx[u][0] + xx[u][1] + yy[u][0] + yy[u][1];
      }
    };
    int b = wave;
    if (b < nblk) load(b, xa, ya);
    for (; b < nblk; b += 2 * WAVES) {
      if (b + WAVES < nblk) load(b + WAVES, xb, yb);
      use(xa, ya);
      if (b + WAVES < nblk) {
        if (b + 2 * WAVES < nblk) load(b + 2 * WAVES, xa, ya);
        use(xb, yb);
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) tot += sxr[k] + srr[k];
    __syncthreads();
  }
  if (tot == 12345.678) out[0] = tot;
}

// arithmetic only: the FMAs of one sweep on register data (no loads), to see the VALU time alone
template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_valu(int N, int iters, double *out) {
  const int rows = N / (WAVES * 64);  // rows per lane
  double tot = 0.0;
  double x = 0.25 + 1e-3 * threadIdx.x, yv = 0.75;
  for (int it = 0; it < iters; it++) {
    double w[K], sxr[K], srr[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      w[k] = 3.0 + 1e-3 * k + 1e-6 * it;
      sxr[k] = srr[k] = 0.0;
    }
    for (int r = 0; r < rows; r += 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
          const double rr = __builtin_fma(-x, w[k], yv);
          sxr[k] = __builtin_fma(x, rr, sxr[k]);
          srr[k] = __builtin_fma(rr, rr, srr[k]);
        }
        asm volatile("" : "+v"(x), "+v"(yv));
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) tot += sxr[k] + srr[k];
    __syncthreads();
  }
  if (tot == 12345.678) out[0] = tot;
}

template <typename F>
static void run(const char *name, F launch, int grid, int iters, int N) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch(grid, 2);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch(grid, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("%-44s grid %4d: %7.2f us/sweep, %6.1f GB/s per WG, %6.2f TB/s total\n", name, grid, us,
         N * 16.0 / us * 1e-3, grid * (N * 16.0) / us * 1e-6);
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000, iters = 200;
  std::vector<double> hx(N), hy(N);
  for (int i = 0; i < N; i++) {
    hx[i] = (i * 2654435761u % 1000003) / 1000003.0 - 0.5;
    hy[i] = 3 * hx[i] + 0.3;
  }
  double *X, *y, *out;
  CK(hipMalloc(&X, N * 8));
  CK(hipMalloc(&y, N * 8));
  CK(hipMalloc(&out, 8));
  CK(hipMemcpy(X, hx.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(y, hy.data(), N * 8, hipMemcpyHostToDevice));
#define DMA(K, WAVES, RING, CHUNK, COMPUTE, grid)                                                            \
  {                                                                                                          \
    auto kern = k_dma<K, WAVES, RING, CHUNK, COMPUTE>;                                                       \
    const size_t dyn = (size_t)WAVES * RING * CHUNK * 16;                                                    \
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                           (int)dyn));                                                                       \
    run("dma K=" #K " waves=" #WAVES " ring=" #RING " chunk=" #CHUNK " compute=" #COMPUTE,                    \
        [&](int g, int it) { hipLaunchKernelGGL(kern, dim3(g), dim3(WAVES * 64), dyn, 0, X, y, N, it, out); }, \
        grid, iters, N);                                                                                     \
  }
#define REG(K, WAVES, UN, COMPUTE, grid) REGX(K, WAVES, UN, COMPUTE, false, true, grid)
#define REGX(K, WAVES, UN, COMPUTE, FMA, NT, grid) REGC(K, WAVES, UN, COMPUTE, FMA, NT, 0, grid)
#define REGC(K, WAVES, UN, COMPUTE, FMA, NT, CACHED, grid)                                                   \
  {                                                                                                          \
    auto kern = k_reg<K, WAVES, UN, COMPUTE, FMA, NT, CACHED>;                                               \
    run("reg K=" #K " waves=" #WAVES " un=" #UN " compute=" #COMPUTE " fma=" #FMA " nt=" #NT " cached%=" #CACHED, \
        [&](int g, int it) { hipLaunchKernelGGL(kern, dim3(g), dim3(WAVES * 64), 0, 0, X, y, N, it, out); }, \
        grid, iters, N);                                                                                     \
  }
  {
    auto kern = k_valu<4, 8>;
    run("valu only K=4 waves=8", [&](int g, int it) { hipLaunchKernelGGL(kern, dim3(g), dim3(512), 0, 0, N, it, out); }, 256, iters, N);
    run("valu only K=4 waves=8", [&](int g, int it) { hipLaunchKernelGGL(kern, dim3(g), dim3(512), 0, 0, N, it, out); }, 32, iters, N);
    auto kern16 = k_valu<4, 16>;
    run("valu only K=4 waves=16", [&](int g, int it) { hipLaunchKernelGGL(kern16, dim3(g), dim3(1024), 0, 0, N, it, out); }, 256, iters, N);
  }
  REGX(4, 8, 4, true, true, false, 256)
  REGC(4, 8, 4, true, true, false, 10, 256)
  REGC(4, 8, 4, true, true, false, 20, 256)
  REGC(4, 8, 4, true, true, false, 30, 256)
  REGC(4, 8, 4, true, true, false, 50, 256)
  REGC(4, 8, 4, true, true, false, 100, 256)
  REGX(4, 8, 4, true, true, false, 32)
  DMA(4, 8, 4, 256, true, 256)
  DMA(4, 8, 4, 256, false, 256)
  DMA(4, 8, 4, 256, true, 128)
  DMA(4, 8, 4, 256, false, 128)
  DMA(4, 8, 4, 256, false, 64)
  DMA(4, 8, 4, 256, false, 32)
  DMA(4, 8, 4, 256, false, 8)
  DMA(8, 8, 4, 256, true, 128)
  DMA(4, 8, 8, 128, true, 256)
  DMA(4, 8, 8, 128, false, 256)
  DMA(4, 8, 2, 512, true, 256)
  DMA(4, 16, 4, 128, true, 256)
  DMA(4, 16, 4, 128, false, 256)
  DMA(4, 4, 4, 512, true, 256)
  REGX(4, 8, 2, true, true, true, 256)
  REGX(4, 8, 2, true, true, false, 256)
  REGX(4, 8, 4, true, true, true, 256)
  REGX(4, 8, 4, true, true, false, 256)
  REGX(4, 8, 1, true, true, false, 256)
  REGX(4, 8, 6, true, true, false, 256)
  REGX(4, 8, 8, true, true, false, 256)
  REGX(4, 8, 8, false, false, false, 256)
  REGX(4, 8, 4, false, false, false, 256)
  REGX(4, 12, 4, true, true, false, 256)
  REGX(4, 16, 4, true, true, false, 256)
  REGX(4, 16, 4, false, false, false, 256)
  REGX(4, 16, 2, true, true, false, 256)
  REGX(4, 4, 4, true, true, false, 256)
  REGX(8, 8, 2, true, true, false, 128)
  REGX(8, 8, 2, true, true, false, 256)
  REGX(4, 8, 2, false, false, false, 256)
  REG(4, 8, 2, true, 256)
  REG(4, 8, 2, false, 256)
  REG(4, 8, 4, true, 256)
  REG(4, 8, 4, false, 256)
  REG(4, 16, 2, true, 256)
  REG(4, 16, 2, false, 256)
  REG(8, 8, 2, true, 128)
  REG(4, 8, 2, false, 128)
  REG(4, 8, 2, false, 32)
  return 0;
}
