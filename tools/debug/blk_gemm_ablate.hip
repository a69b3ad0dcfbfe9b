// Ablation of the in-workgroup product loop (nuts_block.cuh blk_wave_tile): which part keeps the matrix pipe idle?
// MODE 0: MFMAs only (operands in registers); 1: + A fragments from LDS; 2: + B staging through LDS (no global loads);
// 3: + global loads (the real loop); 4: as 2 with a swizzled (conflict-free) staging tile; 5: as 3 swizzled; 6: as 4 without fences.  16 wavefronts per workgroup, NT n-tiles of nk K-tiles each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int P = 4;
template <int MODE>
__global__ __launch_bounds__(1024) void k(const double *Bp, double *out, int Dp, int rep, long long *cycles) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int S = Dp + 2;
  for (int i = threadIdx.x; i < 16 * S + 16 * 288; i += 1024) lds[i] = 1.0;
  __syncthreads();
  double *tb = lds + 16 * S + wave * 288;
  const int fr = lane & 15, fk = lane >> 4, r = lane >> 2, kq = lane & 3;
  const int nk = Dp / 16, NT = Dp / 16;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  d4_t total = {0, 0, 0, 0};
  for (int it = 0; it < rep; it++) {
    for (int nt = wave; nt < NT; nt += 16) {
      const double *pb = Bp + (long long)(nt * 16 + r) * Dp + 2 * kq;
      const double *pa = lds + fr * S + fk;
      d4_t acc = {0, 0, 0, 0};
      d2_t gb[P][2];
#pragma unroll
      for (int s = 0; s < P; s++) {
        gb[s][0] = (MODE == 3 || MODE == 5) ? *reinterpret_cast<const d2_t *>(pb + s * 16) : (d2_t){1.0, 2.0};
        gb[s][1] = (MODE == 3 || MODE == 5) ? *reinterpret_cast<const d2_t *>(pb + s * 16 + 8) : (d2_t){1.0, 2.0};
      }
      for (int kt0 = 0; kt0 + P <= nk; kt0 += P) {
#pragma unroll
        for (int s = 0; s < P; s++) {
          const int kt = kt0 + s;
          if (MODE >= 4) {  // swizzled 16 x 16 tile: column k of row r at k ^ x(r >> 1)
            const int x = (((r >> 1) & 1) << 3) | ((r >> 2) << 1);
            *reinterpret_cast<d2_t *>(&tb[r * 16 + ((2 * kq) ^ x)]) = gb[s][0];
            *reinterpret_cast<d2_t *>(&tb[r * 16 + ((8 + 2 * kq) ^ x)]) = gb[s][1];
          } else if (MODE >= 2) {
            *reinterpret_cast<d2_t *>(&tb[r * 18 + 2 * kq]) = gb[s][0];
            *reinterpret_cast<d2_t *>(&tb[r * 18 + 8 + 2 * kq]) = gb[s][1];
          }
          if (MODE == 3 || MODE == 5) {
            const int kn = kt + P < nk ? kt + P : nk - 1;
            gb[s][0] = *reinterpret_cast<const d2_t *>(pb + kn * 16);
            gb[s][1] = *reinterpret_cast<const d2_t *>(pb + kn * 16 + 8);
          }
          if (MODE != 6) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          double af[4], bf[4];
          const int xr = (((fr >> 1) & 1) << 3) | ((fr >> 2) << 1);
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            af[kk] = MODE >= 1 ? pa[kt * 16 + kk * 4] : (double)(kk + 1);
            bf[kk] = MODE >= 4 ? tb[fr * 16 + ((kk * 4 + fk) ^ xr)]
                               : (MODE >= 2 ? tb[fr * 18 + kk * 4 + fk] : (MODE >= 1 ? 1.5 : (double)(kk + 2)));
          }
#pragma unroll
          for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], acc, 0, 0, 0);
          if (MODE != 6) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      total += acc;
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
  if (total[0] == 12345.678) out[threadIdx.x] = total[1] + total[2] + total[3];
}
template <int MODE>
void run(const double *B, double *out, int Dp, int rep, long long *cyc, int nb) {
  const size_t dyn = (size_t)(16 * (Dp + 2) + 16 * 288) * 8;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pass = 0; pass < 2; pass++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(1024), dyn, 0, B, out, Dp, rep, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> hc(nb * 16);
  hipMemcpy(hc.data(), cyc, nb * 16 * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto v : hc) if ((double)v > mx) mx = (double)v;
  const int NT = Dp / 16, per_simd = ((NT + 15) / 16) * 4 * (Dp / 16 / P * P) * 4;  // MFMAs of the busiest SIMD per product (4 waves)
  printf("Dp=%d MODE %d: %.2f us/product, max-wave ticks/product %.0f, ticks per MFMA of a SIMD %.1f, %.1f TFLOP/s\n", Dp, MODE,
         ms * 1e3 / rep, mx / rep, mx / rep / per_simd, 2.0 * 16 * Dp * Dp * rep * nb / (ms * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
  const int Dp = argc > 1 ? atoi(argv[1]) : 512, rep = 100, nb = 256;
  double *B, *out; long long *cyc;
  hipMalloc(&B, (size_t)Dp * Dp * 8); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, nb * 16 * 8);
  std::vector<double> hb((size_t)Dp * Dp, 0.5);
  hipMemcpy(B, hb.data(), (size_t)Dp * Dp * 8, hipMemcpyHostToDevice);
  run<0>(B, out, Dp, rep, cyc, nb); run<1>(B, out, Dp, rep, cyc, nb); run<2>(B, out, Dp, rep, cyc, nb); run<3>(B, out, Dp, rep, cyc, nb);
  run<4>(B, out, Dp, rep, cyc, nb); run<5>(B, out, Dp, rep, cyc, nb); run<6>(B, out, Dp, rep, cyc, nb);
  return 0;
}
