// How much would more wavefronts in flight help the momentum draw (wave_normals, rng.cuh)?  VERDICT r4 item 4 asks for
// a draw kernel that splits a chain's stream over two wavefronts so that 8 instead of 4 wavefronts share a SIMD.  Before
// building the split this measures its CEILING: the same number of normals from twice / four times as many independent
// streams (no cross-wavefront bookkeeping at all), natural register allocation and capped at 64 VGPRs.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm -I aehmc_amd/csrc \
//        -o /tmp/draw_occupancy_bench tools/debug/experiments/draw_split/draw_occupancy_bench.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rng.cuh"
using namespace aehmc;

template <int WAVES_PER_EU>
__device__ __forceinline__ void body(uint64_t *rng, long long C, long long n, const double *sm, double *out) {
  __shared__ double ztab[ZIG_LDS_DOUBLES];
  const ZigTabLds tab = zig_tab_to_lds(ztab);
  const long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  Pcg64 g = pcg_load(rng + c * 4);
  const PcgLaneJump jump = pcg_lane_jump(g);
  double *dst = out + (size_t)c * n;
#ifdef WITH_PREFETCH_PATCH  /* prefetch_scale.patch applied to csrc/: the scales requested two rounds ahead */
  if (WAVES_PER_EU != -1)
    wave_normals_scaled(g, n, [=](long long i) { return i < n ? sm[i] : 0.0; }, [=](long long i, double z, double s) { dst[i] = s * z; }, tab, jump);
  else
#endif
    wave_normals(g, n, [=](long long i, double z) { dst[i] = sm[i] * z; }, tab, jump);
  if ((threadIdx.x & 63) == 0) pcg_store(rng + c * 4, g);
}
__global__ __launch_bounds__(256) void k_r4(uint64_t *rng, long long C, long long n, const double *sm, double *out) {
  body<-1>(rng, C, n, sm, out);
}
__global__ __launch_bounds__(256) void k_nat(uint64_t *rng, long long C, long long n, const double *sm, double *out) {
  body<0>(rng, C, n, sm, out);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cap64(uint64_t *rng, long long C, long long n,
                                                                                      const double *sm, double *out) {
  body<8>(rng, C, n, sm, out);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_cap80(uint64_t *rng, long long C, long long n,
                                                                                      const double *sm, double *out) {
  body<6>(rng, C, n, sm, out);
}

int main() {
  const long long total = 4096LL * 10000;
  uint64_t *rng; double *sm, *out;
  const long long Cmax = 32768;
  hipMalloc(&rng, Cmax * 4 * 8); hipMalloc(&sm, 10000 * 8); hipMalloc(&out, 2 * total * 8);
  std::vector<uint64_t> h(Cmax * 4);
  for (long long c = 0; c < Cmax; c++) { h[c * 4] = 0x9e3779b97f4a7c15ULL * (c + 1); h[c * 4 + 1] = 0xda942042e4dd58b5ULL * (c + 3);
    h[c * 4 + 2] = c; h[c * 4 + 3] = 2 * c + 1; }
  std::vector<double> hs(10000, 1.0);
  for (int i = 0; i < 10000; i++) hs[i] = 0.5 + 1e-3 * i;
  hipMemcpy(sm, hs.data(), 10000 * 8, hipMemcpyHostToDevice);
  {  // same bits from both forms (normals and final generator states), ragged length
    const long long C = 2048, n = 9973;
    std::vector<double> o0(C * n), o1(C * n);
    std::vector<uint64_t> s0(C * 4), s1(C * 4);
    hipMemcpy(rng, h.data(), Cmax * 4 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_r4, dim3(C / 4), dim3(256), 0, 0, rng, C, n, sm, out);
    hipMemcpy(o0.data(), out, C * n * 8, hipMemcpyDeviceToHost); hipMemcpy(s0.data(), rng, C * 32, hipMemcpyDeviceToHost);
    hipMemcpy(rng, h.data(), Cmax * 4 * 8, hipMemcpyHostToDevice);
    hipMemset(out, 0, C * n * 8);
    hipLaunchKernelGGL(k_nat, dim3(C / 4), dim3(256), 0, 0, rng, C, n, sm, out);
    hipMemcpy(o1.data(), out, C * n * 8, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), rng, C * 32, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (long long i = 0; i < C * n; i++) bad += memcmp(&o0[i], &o1[i], 8) != 0;
    for (long long i = 0; i < C * 4; i++) bad += s0[i] != s1[i];
    printf("prefetching form against round 4's: %lld differing values of %lld\n", bad, C * n + C * 4);
  }
  for (int variant = 0; variant < 3; variant++) {
    for (long long C = 4096; C <= Cmax; C *= 2) {
      const long long n = total / C;
      float best = 1e30f;
      for (int pass = 0; pass < 4; pass++) {
        hipMemcpy(rng, h.data(), Cmax * 4 * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const dim3 grid((unsigned)((C + 3) / 4)), block(256);
        if (variant == 0) hipLaunchKernelGGL(k_nat, grid, block, 0, 0, rng, C, n, sm, out);
        if (variant == 1) hipLaunchKernelGGL(k_cap80, grid, block, 0, 0, rng, C, n, sm, out);
        if (variant == 2) hipLaunchKernelGGL(k_cap64, grid, block, 0, 0, rng, C, n, sm, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const char *nm[3] = {"natural registers", "<= 80 VGPRs (6 waves/SIMD)", "<= 64 VGPRs (8 waves/SIMD)"};
      printf("%-28s  streams %6lld x %6lld normals: %8.1f us  %.3e normals/s  (%lld wavefronts per SIMD offered)\n", nm[variant], C, n,
             best * 1e3, total / (best * 1e-3), C / 1024);
    }
  }
  // the other direction: FEWER wavefronts per SIMD at a fixed stream length -- time that does not grow with the number
  // of resident wavefronts is latency the SIMD could have filled, time that grows in proportion is instruction issue
  for (int variant = 0; variant < 3; variant++)
    for (long long C = 1024; C <= 6144; C += 1024) {
      const long long n = 10000;
      float best = 1e30f;
      for (int pass = 0; pass < 4; pass++) {
        hipMemcpy(rng, h.data(), Cmax * 4 * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const dim3 grid((unsigned)((C + 3) / 4)), block(256);
        if (variant == 0) hipLaunchKernelGGL(k_nat, grid, block, 0, 0, rng, C, n, sm, out);
        if (variant == 1) hipLaunchKernelGGL(k_cap80, grid, block, 0, 0, rng, C, n, sm, out);
        if (variant == 2) hipLaunchKernelGGL(k_r4, grid, block, 0, 0, rng, C, n, sm, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%-28s  %5lld streams x 10000 (%lld wavefronts per SIMD): %8.1f us  %.3e normals/s\n", variant == 2 ? "round 4 (scale loaded at the store)" : variant ? "<= 80 VGPRs" : "natural registers", C,
             C / 1024, best * 1e3, C * n / (best * 1e-3));
    }
  return 0;
}
