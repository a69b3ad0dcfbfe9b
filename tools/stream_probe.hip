// Probe: HBM throughput of the engine's access pattern (one wave per [D]-row, chain-major
// [C,D] fp64) with 8-byte vs 16-byte loads per lane, 3 arrays read + 3 written (a leapfrog).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k8(double* q, double* p, double* g, long long C, long long D, double e) {
  int lane = threadIdx.x & 63; long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return; size_t row = c * D;
  for (long long i = lane; i < D; i += 64) {
    double pp = p[row+i] - e * g[row+i]; double qq = q[row+i] + e * pp; double gg = qq; pp = pp - e * gg;
    q[row+i] = qq; p[row+i] = pp; g[row+i] = gg;
  }
}
__global__ __launch_bounds__(256) void k16(double* q, double* p, double* g, long long C, long long D, double e) {
  int lane = threadIdx.x & 63; long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return; size_t row = c * D;
  for (long long i = 2 * lane; i < D; i += 128) {
    d2 P = *(d2*)(p+row+i), G = *(d2*)(g+row+i), Q = *(d2*)(q+row+i);
    P = P - e * G; Q = Q + e * P; G = Q; P = P - e * G;
    *(d2*)(q+row+i) = Q; *(d2*)(p+row+i) = P; *(d2*)(g+row+i) = G;
  }
}
// one chain per 256-thread block instead of per wave
__global__ __launch_bounds__(256) void k16b(double* q, double* p, double* g, long long C, long long D, double e) {
  long long c = blockIdx.x; if (c >= C) return; size_t row = c * D;
  for (long long i = 2 * threadIdx.x; i < D; i += 512) {
    d2 P = *(d2*)(p+row+i), G = *(d2*)(g+row+i), Q = *(d2*)(q+row+i);
    P = P - e * G; Q = Q + e * P; G = Q; P = P - e * G;
    *(d2*)(q+row+i) = Q; *(d2*)(p+row+i) = P; *(d2*)(g+row+i) = G;
  }
}
int main() {
  long long C = 4096, D = 10000; size_t n = C * D;
  double *q, *p, *g; hipMalloc(&q, n*8); hipMalloc(&p, n*8); hipMalloc(&g, n*8);
  hipMemset(q, 0, n*8); hipMemset(p, 0, n*8); hipMemset(g, 0, n*8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 3; v++) for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    for (int it = 0; it < 20; it++) {
      if (v == 0) hipLaunchKernelGGL(k8, dim3(C/4), dim3(256), 0, 0, q, p, g, C, D, 1e-3);
      if (v == 1) hipLaunchKernelGGL(k16, dim3(C/4), dim3(256), 0, 0, q, p, g, C, D, 1e-3);
      if (v == 2) hipLaunchKernelGGL(k16b, dim3(C), dim3(256), 0, 0, q, p, g, C, D, 1e-3);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d (%s): %.3f ms/launch  %.0f GB/s\n", v, v==0?"8B/lane wave-per-row":v==1?"16B/lane wave-per-row":"16B/lane block-per-row", ms/20, 6.0*n*8/(ms/20*1e-3)/1e9);
  }
  return 0;
}
