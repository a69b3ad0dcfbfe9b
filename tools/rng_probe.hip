// Where does a round of wave_normals spend its time?  Variants of the 64-wide round timed on
// 4096 chains x 10000 draws.  build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/rng_probe tools/rng_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../aehmc_amd/csrc/rng.cuh"
using namespace aehmc;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)


// copy of wave_normals whose rejection handler is a stand-in: MODE 1 = consume one raw output and
// accept (walk overhead only); MODE 2 = one exp() on the VALU; MODE 3 = real handler
template <int MODE, class Store, class Tab>
__device__ inline void wave_normals_v(Pcg64 &rng, long long n, Store store, const Tab &tab) {
  const int lane = threadIdx.x & 63;
  const u128 Ak = (((u128)c_pcg_jump[lane][0]) << 64) | (u128)c_pcg_jump[lane][1];
  const u128 GI = ((((u128)c_pcg_jump[lane][2]) << 64) | (u128)c_pcg_jump[lane][3]) * rng.inc;
  long long pos = 0;
  while (pos < n) {
    const u128 sk = Ak * rng.state + GI;
    const uint64_t raw = pcg_output(sk);
    const ZigDraw d = zig_fast(raw, tab);
    const unsigned long long fail = __ballot(!d.accept);
    int cur = 0;
    for (;;) {
      const unsigned long long m = cur < 64 ? (fail & (~0ULL << cur)) : 0ULL;
      const int f = m ? (__ffsll((long long)m) - 1) : 64;
      const long long want = n - pos;
      if ((long long)(f - cur) >= want) {
        if (lane >= cur && lane < cur + want) store(pos + (lane - cur), d.x);
        rng.state = shfl_u128(sk, cur + (int)want - 1);
        return;
      }
      if (lane >= cur && lane < f) store(pos + (lane - cur), d.x);
      pos += f - cur;
      if (f == 64) { rng.state = shfl_u128(sk, 63); break; }
      ZigDraw df;
      df.x = __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(d.x), f));
      df.rabs = shfl_u64(d.rabs, f);
      df.idx = __builtin_amdgcn_readlane(d.idx, f);
      df.accept = false;
      LaneSrc src{raw, f + 1};
      double z = df.x;
      bool ok = true;
      if (MODE == 1) { uint64_t r; ok = src.next(r); }
      if (MODE == 2) { uint64_t r; ok = src.next(r); z = exp(-0.5 * df.x * df.x) * u64_to_unit(r); }
      if (MODE == 3) ok = zig_slow_from(src, df, z);
      if (ok) {
        if (lane == 0) store(pos, z);
        pos++;
        cur = src.idx;
        if (pos == n) { rng.state = shfl_u128(sk, cur - 1); return; }
      } else {
        rng.state = shfl_u128(sk, f);
        if (MODE == 3) z = zig_slow(rng, df); else pcg_next64(rng);
        if (lane == 0) store(pos, z);
        pos++;
        break;
      }
    }
  }
}

// V=0 full (LDS tables) ; V=1 no stores (checksum) ; V=2 fast path only, failures ignored ;
// V=3 only the LCG jump + output (no ziggurat) ; V=4 full with constant-memory tables
template <int V>
__global__ __launch_bounds__(256) void k(uint64_t *rng, long long C, long long n, double *out) {
  __shared__ double ztab[512];
  const ZigTabLds tab = zig_tab_to_lds(ztab);
  const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const int lane = threadIdx.x & 63;
  Pcg64 g = pcg_load(rng + c * 4);
  if (V == 0) wave_normals(g, n, [=](long long i, double z) { out[c * n + i] = z; }, tab);
  if (V == 4) wave_normals(g, n, [=](long long i, double z) { out[c * n + i] = z; });
  if (V == 5) wave_normals_v<1>(g, n, [=](long long i, double z) { out[c * n + i] = z; }, tab);
  if (V == 6) wave_normals_v<2>(g, n, [=](long long i, double z) { out[c * n + i] = z; }, tab);
  if (V == 7) wave_normals_v<3>(g, n, [=](long long i, double z) { out[c * n + i] = z; }, tab);
  if (V == 1) {
    double acc = 0.0;
    wave_normals(g, n, [&](long long i, double z) { acc += z; }, tab);
    if (acc == 1.2345) out[c * n + lane] = acc;
  }
  if (V == 2 || V == 3) {
    const u128 Ak = (((u128)c_pcg_jump[lane][0]) << 64) | (u128)c_pcg_jump[lane][1];
    const u128 GI = ((((u128)c_pcg_jump[lane][2]) << 64) | (u128)c_pcg_jump[lane][3]) * g.inc;
    double acc = 0.0;
    for (long long pos = 0; pos < n; pos += 64) {
      const u128 sk = Ak * g.state + GI;
      const uint64_t raw = pcg_output(sk);
      if (V == 2) {
        const ZigDraw d = zig_fast(raw, tab);
        out[c * n + pos + lane] = d.x;
        if (__ballot(!d.accept) == 0x123456789ULL) acc += 1.0;
      } else {
        acc += (double)(raw >> 40);
      }
      g.state = shfl_u128(sk, 63);
    }
    if (acc == 1.2345) out[c * n + lane] = acc;
  }
  if (lane == 0) pcg_store(rng + c * 4, g);
}

template <int V>
int run(uint64_t *rng, long long C, long long n, double *out, const char *what) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<V>, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, rng, C, n, out);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, 0, rng, C, n, out);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %8.1f us/launch  %6.0f clk/round/wave@2.4GHz(4 waves/SIMD)\n", what, ms / 5 * 1e3,
         ms / 5 * 1e-3 / ((n + 63) / 64) * 2.4e9 / 4);
  return 0;
}

int main() {
  const long long C = 4096, n = 10000;
  uint64_t jump[64][4];
  u128 A = 1, G = 0;
  for (int k2 = 0; k2 < 64; k2++) {
    G = G * AEHMC_PCG_MULT + 1; A = A * AEHMC_PCG_MULT;
    jump[k2][0] = (uint64_t)(A >> 64); jump[k2][1] = (uint64_t)A; jump[k2][2] = (uint64_t)(G >> 64); jump[k2][3] = (uint64_t)G;
  }
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_pcg_jump), jump, sizeof(jump)));
  std::vector<uint64_t> h(C * 4);
  for (long long i = 0; i < C; i++) { h[4*i] = 0x1234567 + i; h[4*i+1] = 0x9e3779b97f4a7c15ULL * (i + 1); h[4*i+2] = 77 + i; h[4*i+3] = (2 * i + 1); }
  uint64_t *rng; double *out;
  CHK(hipMalloc(&rng, C * 32)); CHK(hipMalloc(&out, C * n * 8));
  CHK(hipMemcpy(rng, h.data(), C * 32, hipMemcpyHostToDevice));
  if (run<0>(rng, C, n, out, "full, LDS tables")) return 1;
  if (run<4>(rng, C, n, out, "full, constant-memory tables")) return 1;
  if (run<1>(rng, C, n, out, "full, no stores")) return 1;
  if (run<5>(rng, C, n, out, "walk only: rejection = consume 1, accept")) return 1;
  if (run<6>(rng, C, n, out, "walk + one exp() per rejection")) return 1;
  if (run<7>(rng, C, n, out, "copy with the real handler")) return 1;
  if (run<2>(rng, C, n, out, "fast path only (failures ignored), stores")) return 1;
  if (run<3>(rng, C, n, out, "LCG jump + output only")) return 1;
  return 0;
}
