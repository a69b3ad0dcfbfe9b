// Microbenchmark: sustained v_fma_f64 rate on MI355X (no memory traffic), the ceiling of the fp64
// VALU-bound kernels (k_hmc_fused at c2, k_nuts_linreg at c5, k_nuts_wide).  The datasheet's 78.6
// TFLOP/s is 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz; under sustained fp64 vector load the
// chip does not hold 2.4 GHz.  Reports TFLOP/s for the whole chip and for a launch that keeps only
// 32 / 64 / 128 CUs busy (power-limited clocks show as a higher per-CU rate there).
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/valu_f64_peak.hip -o tools/bin/valu_f64_peak && tools/bin/valu_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 1e-3 * i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters / 8; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    asm volatile("" : "+v"(a), "+v"(b));
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int grid, int reps) {
  const int iters = 40000;
  double *out;
  hipMalloc(&out, sizeof(double) * grid * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0 - 1e-9, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0 - 1e-9, 0.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)reps * grid * 256.0 * iters * NACC;
  // a workgroup of 4 waves occupies one CU (one wave per SIMD) when grid <= 256; more workgroups stack up
  const double cus = grid < 256 ? grid : 256;
  printf("NACC=%2d grid=%5d : %6.2f TFLOP/s  (%7.1f ms)  = %.2f GHz-equivalent at 16 lanes x 2 flop per SIMD-cycle on %3.0f CUs\n",
         NACC, grid, 2 * fma / ms / 1e9, ms, 2 * fma / (ms * 1e-3) / (cus * 4 * 16 * 2) / 1e9, cus);
  hipFree(out);
}
int main() {
  run<8>(256, 4);       // 1 wave per SIMD, 8 independent chains
  run<16>(256, 4);
  run<8>(1024, 4);      // 4 waves per SIMD
  run<8>(2048, 40);     // 8 waves per SIMD, ~ a second of sustained load: DVFS-settled rate
  run<8>(32, 4);        // an eighth of the chip busy
  run<8>(64, 4);
  run<8>(128, 4);
  return 0;
}
