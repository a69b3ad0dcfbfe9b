// Microbenchmark of the in-workgroup LDS -> LDS product: round 4's 16 wavefronts per workgroup (nuts_block_reg.cuh)
// against the 4 wavefronts of k_nuts_block_flow (plain / software-pipelined); prints us per product + barrier.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm -I aehmc_amd/csrc -o /tmp/blk_gemm4_bench tools/debug/blk_gemm4_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nuts_block_flow.cuh"
using namespace aehmc;

template <int VARIANT>
__global__ __launch_bounds__(VARIANT == 0 ? BLK_THREADS : BQ_THREADS) void k_bench(const double *B, double *out, long long D, int rep) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int S = (int)blk_lds_stride(D);
  double *xbuf = lds, *ybuf = lds + BLK_CHAINS * S;
  double *tb = lds + 2 * BLK_CHAINS * S + wave * (VARIANT == 0 ? BLK_TB : BQ_TILES * BLK_TB);
  for (int k = threadIdx.x; k < 2 * BLK_CHAINS * S; k += blockDim.x) lds[k] = (k % S) < D ? 1e-3 * (k % 7) : 0.0;
  __syncthreads();
  for (int r = 0; r < rep; r++) {
    if (VARIANT == 0) blk_gemm_lds(xbuf, ybuf, S, B, D, wave, lane, tb);
    if (VARIANT == 1) blk_gemm_lds4<0>(xbuf, ybuf, S, B, D, wave, lane, tb);
    if (VARIANT == 2) blk_gemm_lds4<1>(xbuf, ybuf, S, B, D, wave, lane, tb);
    blk_barrier_lds();
    double *t = xbuf; xbuf = ybuf; ybuf = t;
  }
  if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = xbuf[threadIdx.x * S];
}

int main(int argc, char **argv) {
  const long long D = argc > 1 ? atoll(argv[1]) : 200;
  const int rep = argc > 2 ? atoi(argv[2]) : 400, nb = 256;
  const long long Dp = (D + 15) / 16 * 16;
  double *B, *out;
  hipMalloc(&B, Dp * Dp * 8); hipMalloc(&out, nb * 16 * 8);
  std::vector<double> hb(Dp * Dp, 0.0);
  for (long long i = 0; i < D; i++) for (long long j = 0; j < D; j++) hb[i * Dp + j] = (i == j) ? 0.5 : 1e-3;
  hipMemcpy(B, hb.data(), Dp * Dp * 8, hipMemcpyHostToDevice);
  const size_t dyn = blk_flow_lds_bytes(D);
  const char *names[3] = {"16 wavefronts (round 4)", "4 wavefronts, plain", "4 wavefronts, pipelined"};
  for (int variant = 0; variant < 3; variant++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int pass = 0; pass < 2; pass++) {
      hipEventRecord(e0);
      if (variant == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((k_bench<0>), dim3(nb), dim3(BLK_THREADS), dyn, 0, B, out, D, rep);
      } else if (variant == 1) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((k_bench<1>), dim3(nb), dim3(BQ_THREADS), dyn, 0, B, out, D, rep);
      } else {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((k_bench<2>), dim3(nb), dim3(BQ_THREADS), dyn, 0, B, out, D, rep);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 2.0 * 16 * Dp * Dp * rep * nb;
    printf("D=%lld %-26s %.2f us / product + barrier, %.1f TFLOP/s (padded), prefetch %d\n", D, names[variant], ms * 1e3 / rep,
           flops / (ms * 1e-3) / 1e12, BLK_PREFETCH);
  }
  return 0;
}
