// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (no memory traffic).
// Calibrates the fp64 MFMA roofline used by bench.py (the microarch guide has no f64 row).
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters / 16; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int seconds_hint) {
  int iters = 20000;
  int grid = 256 * blocks_per_cu;
  double *out;
  hipMalloc(&out, sizeof(double) * grid * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    for (int r = 0; r < seconds_hint; r++) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)seconds_hint * grid * 4.0 * iters * NACC * 2048.0;
    printf("NACC=%d blocks/CU=%d : %.2f TFLOP/s  (%.1f ms)  cycles/MFMA/SIMD@2.4GHz=%.1f\n", NACC, blocks_per_cu,
           flops / ms / 1e9, ms, 2.4e9 * (ms / 1e3) / ((double)seconds_hint * iters * NACC * blocks_per_cu));
  }
  hipFree(out);
}
int main() {
  run<4>(1, 4);
  run<8>(1, 4);
  run<16>(1, 2);
  run<8>(2, 2);
  run<16>(2, 20);  // ~ seconds of sustained load: DVFS-settled rate
  return 0;
}
