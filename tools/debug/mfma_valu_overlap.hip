// Do fp64 MFMAs of one wavefront overlap with fp64 / integer VALU work of ANOTHER wavefront on the same SIMD (gfx950)?
// One workgroup of 512 threads: waves 0-3 (one per SIMD) run a chain of v_mfma_f64_16x16x4_f64, waves 4-7 (their SIMD
// partners) a chain of VALU instructions of one kind.  Cycles of the MFMA waves alone, of the VALU waves alone, of both.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/mvo tools/debug/mfma_valu_overlap.hip ; run: /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int n_mfma, int n_valu, int mode, long long *out, double *sink, int prio) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
  __syncthreads();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  if (do_mfma) {
    d4_t acc = {0, 0, 0, 0};
    double af = 1.0 + lane * 1e-3, bf = 1.0 - lane * 1e-3;
    for (int i = 0; i < n_mfma; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  }
  if (do_valu) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    if (KIND == 3) {  // ONE dependent fp64 FMA chain: latency-bound, the issue port is mostly free
      double x0 = lane;
      const double a = 1.0000001, b = 1e-9;
      for (int i = 0; i < n_valu; i += 4) {
        x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b); x0 = __builtin_fma(x0, a, b);
      }
      sink[threadIdx.x] = x0;
    } else if (KIND == 0) {  // dependent fp64 FMA chain (4 independent chains)
      double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      const double a = 1.0000001, b = 1e-9;
      for (int i = 0; i < n_valu; i += 4) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
      }
      sink[threadIdx.x] = x0 + x1 + x2 + x3;
    } else if (KIND == 1) {  // 32-bit integer multiplies (the PCG step's 128-bit product is made of these)
      unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      for (int i = 0; i < n_valu; i += 4) {
        x0 = x0 * 2654435761u + 1u; x1 = x1 * 2654435761u + 1u; x2 = x2 * 2654435761u + 1u; x3 = x3 * 2654435761u + 1u;
      }
      sink[threadIdx.x] = (double)(x0 + x1 + x2 + x3);
    } else {  // fp32 FMA
      float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      const float a = 1.0000001f, b = 1e-9f;
      for (int i = 0; i < n_valu; i += 4) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      }
      sink[threadIdx.x] = x0 + x1 + x2 + x3;
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
}

template <int KIND>
void run(const char *name, int n_mfma, int n_valu, int prio = 0) {
  long long *out; double *sink;
  hipMalloc(&out, 8 * sizeof(long long)); hipMalloc(&sink, 512 * sizeof(double));
  long long h[8];
  printf("%s%s: %d MFMAs on waves 0-3 | %d VALU instructions on waves 4-7 (s_memtime ticks)\n", name, prio ? " (VALU waves at s_setprio 3)" : "", n_mfma, n_valu);
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(512), 0, 0, n_mfma, n_valu, mode, out, sink, prio);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("  mode %s: mfma wave %lld  valu wave %lld\n", mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both     ", h[0], h[4]);
  }
  hipFree(out); hipFree(sink);
}
int main() {
  run<0>("fp64 FMA ", 4000, 64000);
  run<1>("int32 mul", 4000, 64000);
  run<2>("fp32 FMA ", 4000, 64000);
  run<3>("one dependent fp64 FMA chain", 4000, 16000);
  run<3>("one dependent fp64 FMA chain", 4000, 16000, 1);
  run<0>("fp64 FMA ", 4000, 64000, 1);
  run<1>("int32 mul", 4000, 64000, 1);
  return 0;
}
