// Microbenchmark of the in-workgroup product of nuts_block.cuh: 256 workgroups x 16 wavefronts, each workgroup
// repeats out[16][D] = A[16][D] * B[D][D]^T (A rows in LDS, B from L2) REP times; prints cycles per product.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I aehmc_amd/csrc -o /tmp/blk_gemm_bench tools/debug/blk_gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nuts_block.cuh"
using namespace aehmc;

template <int VARIANT>
__global__ __launch_bounds__(BLK_THREADS) void k_bench(const double *X, const double *B, double *out, long long D, long long C,
                                                        int rep, long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS;
  const int S = (int)blk_lds_stride(D);
  BlkTimer tm;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int r = 0; r < rep; r++) {
    if (VARIANT == 0) blk_gemm(lds, S, X, B, out, D, c0, C, wave, lane, tm);
    if (VARIANT == 1) {  // MFMA tiles only (operand rows staged once)
      if (r == 0) {
        const long long c = c0 + wave;
        for (int k = lane; k < S; k += 64) lds[wave * S + k] = (c < C && k < D) ? X[c * D + k] : 0.0;
        __syncthreads();
      }
      const int NT = (int)((D + 15) / 16);
      double *const tb = lds + BLK_CHAINS * S + wave * BLK_TB;
      for (int nt = wave; nt < NT; nt += BLK_CHAINS)
        blk_wave_tile(lds, S, B, NT * 16, D, nt * 16, out + c0 * D, D, 0xffffu, lane, tb);
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

int main(int argc, char **argv) {
  const long long D = argc > 1 ? atoll(argv[1]) : 200, C = argc > 2 ? atoll(argv[2]) : 4096;
  const int rep = argc > 3 ? atoi(argv[3]) : 200;
  double *X, *B, *out;
  long long *cyc;
  hipMalloc(&X, C * D * 8); hipMalloc(&B, (D + 16) * (D + 16) * 8); hipMalloc(&out, C * D * 8);
  const int nb = (int)((C + 15) / 16);
  hipMalloc(&cyc, nb * 16 * 8);
  std::vector<double> h(C * D, 1.0), hb((D + 16) * (D + 16), 0.5);
  hipMemcpy(X, h.data(), C * D * 8, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), (D + 16) * (D + 16) * 8, hipMemcpyHostToDevice);
  const size_t dyn = blk_lds_bytes(D) + BLK_CHAINS * BLK_TB * 8;
  for (int variant = 0; variant < 2; variant++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++) {
      hipEventRecord(e0);
      if (variant == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((k_bench<0>), dim3(nb), dim3(BLK_THREADS), dyn, 0, X, B, out, D, C, rep, cyc);
      } else {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bench<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((k_bench<1>), dim3(nb), dim3(BLK_THREADS), dyn, 0, X, B, out, D, C, rep, cyc);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(nb * 16);
    hipMemcpy(hc.data(), cyc, nb * 16 * 8, hipMemcpyDeviceToHost);
    double mx = 0, mean = 0;
    for (auto v : hc) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= hc.size();
    const double flops = 2.0 * 16 * D * D * rep * nb;
    printf("D=%lld C=%lld variant %d (%s): %.3f ms for %d products = %.2f us/product; %.0f cycles/product (max wave %.0f); %.1f TFLOP/s; MFMA-bound floor %.0f cycles\n",
           D, C, variant, variant ? "tiles only" : "stage + barriers + tiles", ms, rep, ms * 1e3 / rep, mean / rep, mx / rep,
           flops / (ms * 1e-3) / 1e12, (double)(((D + 15) / 16 + 3) / 4) * ((D + 15) / 16) * 4 * 64);
  }
  return 0;
}
