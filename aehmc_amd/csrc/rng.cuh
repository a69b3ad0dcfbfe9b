// Device RNG for the many-chain HMC/NUTS engine (gfx950).
//
// Restates, for HIP, what the reference's RandomStream call sites execute through
// aesara -> numpy (SURVEY.md 8c "scheme A"): one PCG64 per call site per chain,
//   srng.normal    -> Generator.normal(0,1)      -> random_standard_normal (256-layer ziggurat)
//   srng.bernoulli -> Generator.binomial(1, p)   -> random_binomial_inversion with n == 1
// (aehmc/metrics.py:66, trajectory.py:516, proposals.py:99,131, hmc.py:194).
// All functions here are wave-uniform unless they say "per lane": every lane of the
// 64-wide wavefront that owns a chain carries the same generator state in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aehmc_ziggurat_tables.h"

namespace aehmc {

typedef unsigned __int128 u128;

__constant__ uint64_t c_zig_ki[256] = {AEHMC_ZIG_KI_VALUES};
__constant__ double c_zig_wi[256] = {AEHMC_ZIG_WI_VALUES};
__constant__ double c_zig_fi[256] = {AEHMC_ZIG_FI_VALUES};
// c_pcg_jump[k] = {A^(k+1) hi, lo, G_(k+1) hi, lo} with A the PCG64 multiplier and
// G_n = 1 + A + ... + A^(n-1):  state_{t+n} = A^n * state_t + G_n * inc   (mod 2^128).
// Filled by aehmc_create().
__constant__ uint64_t c_pcg_jump[64][4];

#define AEHMC_PCG_MULT ((((u128)2549297995355413924ULL) << 64) | (u128)4865540595714422341ULL)

struct Pcg64 {
  u128 state, inc;
};

__device__ __forceinline__ Pcg64 pcg_load(const uint64_t *s) {
  Pcg64 r;
  r.state = (((u128)s[0]) << 64) | (u128)s[1];
  r.inc = (((u128)s[2]) << 64) | (u128)s[3];
  return r;
}
__device__ __forceinline__ void pcg_store(uint64_t *s, const Pcg64 &r) {
  s[0] = (uint64_t)(r.state >> 64);
  s[1] = (uint64_t)r.state;
}
__device__ __forceinline__ uint64_t pcg_output(u128 s) {  // XSL-RR
  uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
__device__ __forceinline__ uint64_t pcg_next64(Pcg64 &r) {
  r.state = r.state * AEHMC_PCG_MULT + r.inc;
  return pcg_output(r.state);
}
__device__ __forceinline__ double pcg_next_double(Pcg64 &r) {
  return (double)(pcg_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- ziggurat -------------------------------------------------------------------
struct ZigDraw {
  double x;
  uint64_t rabs;
  int idx;
  bool accept;
};
__device__ __forceinline__ ZigDraw zig_fast(uint64_t r) {
  ZigDraw d;
  d.idx = (int)(r & 0xff);
  r >>= 8;
  int sign = (int)(r & 0x1);
  d.rabs = (r >> 1) & 0x000fffffffffffffULL;
  double x = (double)d.rabs * c_zig_wi[d.idx];
  d.x = sign ? -x : x;
  d.accept = d.rabs < c_zig_ki[d.idx];
  return d;
}
// everything after a failed fast-path test of draw `d` (tail / wedge / full redraws)
__device__ inline double zig_slow(Pcg64 &rng, ZigDraw d) {
  for (;;) {
    if (d.idx == 0) {
      for (;;) {
        double xx = -AEHMC_ZIG_NOR_INV_R * log1p(-pcg_next_double(rng));
        double yy = -log1p(-pcg_next_double(rng));
        if (yy + yy > xx * xx)
          return ((d.rabs >> 8) & 0x1) ? -(AEHMC_ZIG_NOR_R + xx) : AEHMC_ZIG_NOR_R + xx;
      }
    } else {
      if (((c_zig_fi[d.idx - 1] - c_zig_fi[d.idx]) * pcg_next_double(rng) + c_zig_fi[d.idx]) <
          exp(-0.5 * d.x * d.x))
        return d.x;
    }
    d = zig_fast(pcg_next64(rng));
    if (d.accept) return d.x;
  }
}
__device__ inline double rng_standard_normal(Pcg64 &rng) {
  ZigDraw d = zig_fast(pcg_next64(rng));
  return d.accept ? d.x : zig_slow(rng, d);
}

// ---- binomial(1, p) --------------------------------------------------------------
__device__ inline int binomial1_inversion(Pcg64 &rng, double p) {
  double q = 1.0 - p;
  double qn = exp(1 * log(q));
  // numpy: bound = min(n, n p + 10 sqrt(n p q + 1)); with n == 1 the second term is >= 10
  const long long bound = 1;
  long long X = 0;
  double px = qn;
  double U = pcg_next_double(rng);
  while (U > px) {
    X++;
    if (X > bound) {
      X = 0;
      px = qn;
      U = pcg_next_double(rng);
    } else {
      U -= px;
      px = ((1 - X + 1) * p * px) / (X * q);
    }
  }
  return (int)X;
}
__device__ inline int rng_bernoulli(Pcg64 &rng, double p) {
  if (p == 0.0) return 0;  // numpy draws nothing
  if (p <= 0.5) return binomial1_inversion(rng, p);
  return 1 - binomial1_inversion(rng, 1.0 - p);
}

// ---- wave-cooperative sequence of normals ----------------------------------------
// broadcast lane `src` (wave-uniform) to the whole wave: v_readlane, result in SGPRs
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return (((uint64_t)hi) << 32) | lo;
}
__device__ __forceinline__ u128 shfl_u128(u128 v, int src) {
  return (((u128)shfl_u64((uint64_t)(v >> 64), src)) << 64) | (u128)shfl_u64((uint64_t)v, src);
}

// Draws z_0..z_{n-1} exactly as n sequential Generator.normal() calls would, with the
// 64 lanes of the wave evaluating 64 consecutive stream positions at once (LCG
// jump-ahead); a rejected fast-path test (0.7 % of draws) is resolved serially and the
// wave restarts behind it.  `store(i, z)` is called by exactly one lane per element.
template <class Store>
__device__ inline void wave_normals(Pcg64 &rng, long long n, Store store) {
  const int lane = threadIdx.x & 63;
  const u128 Ak = (((u128)c_pcg_jump[lane][0]) << 64) | (u128)c_pcg_jump[lane][1];
  const u128 Gk = (((u128)c_pcg_jump[lane][2]) << 64) | (u128)c_pcg_jump[lane][3];
  long long pos = 0;
  while (pos < n) {
    long long rem = n - pos;
    int need = rem < 64 ? (int)rem : 64;
    u128 sk = Ak * rng.state + Gk * rng.inc;  // state after lane+1 steps
    ZigDraw d = zig_fast(pcg_output(sk));
    unsigned long long fail = __ballot(!d.accept && lane < need);
    int f = fail ? (__ffsll((long long)fail) - 1) : need;
    if (lane < f) store(pos + lane, d.x);
    if (f < need) {
      ZigDraw df;
      df.x = __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(d.x), f));
      df.rabs = shfl_u64(d.rabs, f);
      df.idx = __builtin_amdgcn_readlane(d.idx, f);
      df.accept = false;
      rng.state = shfl_u128(sk, f);
      double z = zig_slow(rng, df);
      if (lane == 0) store(pos + f, z);
      pos += f + 1;
    } else {
      rng.state = shfl_u128(sk, need - 1);
      pos += need;
    }
  }
}

}  // namespace aehmc
