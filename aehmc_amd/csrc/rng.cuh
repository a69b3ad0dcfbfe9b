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
#ifndef __HIPCC_RTC__  /* (hipRTC supplies the runtime, the math functions and the fixed-width integers itself) */
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "../../include/aehmc_ziggurat_tables.h"

namespace aehmc {

// 128-bit unsigned integers of the PCG64 state as two 64-bit halves with explicit arithmetic.  (With
// `unsigned __int128` state hipcc 7.2 lost the upper half of the loop-carried generator state inside the
// inlined ziggurat redraw loop of k_nuts_resident<1,1>: the registers of an exp() coefficient were
// multiplied in its place, so the redraws ran on a garbage state.  The RNG-state parity tests cover
// every kernel family that draws normals, at sizes that take the redraw path.)
struct u128 {
  uint64_t hi, lo;
};
__host__ __device__ __forceinline__ u128 mk128(uint64_t hi, uint64_t lo) {
  u128 r;
  r.hi = hi;
  r.lo = lo;
  return r;
}
// The arithmetic is written on the 64-bit halves too (round 3): the low 128 bits of a product are
//   lo = a.lo * b.lo (low word),  hi = umulhi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi   (mod 2^64),
// and a sum carries with a compare -- no `__int128` value or operation exists in the IR, so the pass that
// lost the upper word has nothing to work on, whatever instantiation or compiler comes next.
__device__ __forceinline__ u128 mul128(u128 a, u128 b) {  // low 128 bits
  return mk128(__umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi, a.lo * b.lo);
}
__device__ __forceinline__ u128 add128(u128 a, u128 b) {
  const uint64_t lo = a.lo + b.lo;
  return mk128(a.hi + b.hi + (lo < a.lo ? 1ULL : 0ULL), lo);
}
__device__ __forceinline__ u128 muladd128(u128 a, u128 b, u128 c) { return add128(mul128(a, b), c); }

// One copy per translation unit (the library is several: tu_*.hip) -- internal linkage in the ahead-of-time build;
// hipRTC programs are single translation units and address the tables by name.
#ifdef __HIPCC_RTC__
#define AEHMC_TU_LOCAL
#else
#define AEHMC_TU_LOCAL static
#endif
AEHMC_TU_LOCAL __constant__ uint64_t c_zig_ki[256] = {AEHMC_ZIG_KI_VALUES};
AEHMC_TU_LOCAL __constant__ double c_zig_wi[256] = {AEHMC_ZIG_WI_VALUES};
AEHMC_TU_LOCAL __constant__ double c_zig_fi[256] = {AEHMC_ZIG_FI_VALUES};

#define AEHMC_PCG_MULT_HI 2549297995355413924ULL
#define AEHMC_PCG_MULT_LO 4865540595714422341ULL

// c_pcg_jump[k] = {A^(k+1) hi, lo, G_(k+1) hi, lo} with A the PCG64 multiplier and
// G_n = 1 + A + ... + A^(n-1):  state_{t+n} = A^n * state_t + G_n * inc   (mod 2^128).
// A compile-time table (round 5; until then aehmc_create() filled it in every code object at run time): 128-bit
// products from 32-bit limbs so that the constant evaluator needs no `__int128`.
struct PcgJumpTable {
  uint64_t v[64][4];
};
constexpr uint64_t cx_mulhi64(uint64_t a, uint64_t b) {
  const uint64_t a0 = a & 0xffffffffULL, a1 = a >> 32, b0 = b & 0xffffffffULL, b1 = b >> 32;
  const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const uint64_t mid = (p00 >> 32) + (p01 & 0xffffffffULL) + (p10 & 0xffffffffULL);
  return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}
constexpr PcgJumpTable make_pcg_jump_table() {
  PcgJumpTable t{};
  uint64_t Ah = 0, Al = 1, Gh = 0, Gl = 0;
  for (int k = 0; k < 64; k++) {
    // G_{k+1} = G_k * mult + 1,  A_{k+1} = A_k * mult   (low 128 bits)
    uint64_t h = cx_mulhi64(Gl, AEHMC_PCG_MULT_LO) + Gh * AEHMC_PCG_MULT_LO + Gl * AEHMC_PCG_MULT_HI;
    uint64_t l = Gl * AEHMC_PCG_MULT_LO;
    Gl = l + 1;
    Gh = h + (Gl < l ? 1ULL : 0ULL);
    h = cx_mulhi64(Al, AEHMC_PCG_MULT_LO) + Ah * AEHMC_PCG_MULT_LO + Al * AEHMC_PCG_MULT_HI;
    Al = Al * AEHMC_PCG_MULT_LO;
    Ah = h;
    t.v[k][0] = Ah; t.v[k][1] = Al; t.v[k][2] = Gh; t.v[k][3] = Gl;
  }
  return t;
}
AEHMC_TU_LOCAL __constant__ PcgJumpTable c_pcg_jump_table = make_pcg_jump_table();
#define c_pcg_jump c_pcg_jump_table.v

struct Pcg64 {
  u128 state, inc;
};

__device__ __forceinline__ Pcg64 pcg_load(const uint64_t *s) {
  Pcg64 r;
  r.state = mk128(s[0], s[1]);
  r.inc = mk128(s[2], s[3]);
  return r;
}
__device__ __forceinline__ void pcg_store(uint64_t *s, const Pcg64 &r) {
  s[0] = r.state.hi;
  s[1] = r.state.lo;
}
__device__ __forceinline__ uint64_t pcg_output(u128 s) {  // XSL-RR
  uint64_t hi = s.hi, lo = s.lo;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
__device__ __forceinline__ uint64_t pcg_next64(Pcg64 &r) {
  r.state = muladd128(r.state, mk128(AEHMC_PCG_MULT_HI, AEHMC_PCG_MULT_LO), r.inc);
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
// where the fast path reads its two tables: constant memory (wave-uniform index -> scalar
// loads) or a copy in LDS (per-lane index: ~5x shorter latency than the vector-memory path)
constexpr int ZIG_LDS_DOUBLES = 768;  // wi, ki, fi: 256 entries each
struct ZigTabConst {
  __device__ __forceinline__ double wi(int i) const { return c_zig_wi[i]; }
  __device__ __forceinline__ uint64_t ki(int i) const { return c_zig_ki[i]; }
  __device__ __forceinline__ double fi(int i) const { return c_zig_fi[i]; }
};
struct ZigTabLds {
  const double *w;
  const uint64_t *k;
  const double *f;
  __device__ __forceinline__ double wi(int i) const { return w[i]; }
  __device__ __forceinline__ uint64_t ki(int i) const { return k[i]; }
  __device__ __forceinline__ double fi(int i) const { return f[i]; }
};
// copies the tables into `lds` (ZIG_LDS_DOUBLES x 8 B); every thread of the block must call it
__device__ __forceinline__ ZigTabLds zig_tab_to_lds(double *lds) {
  uint64_t *k = reinterpret_cast<uint64_t *>(lds + 256);
  double *f = lds + 512;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    lds[i] = c_zig_wi[i];
    k[i] = c_zig_ki[i];
    f[i] = c_zig_fi[i];
  }
  __syncthreads();
  return ZigTabLds{lds, k, f};
}
template <class Tab>
__device__ __forceinline__ ZigDraw zig_fast(uint64_t r, const Tab &tab) {
  ZigDraw d;
  d.idx = (int)(r & 0xff);
  r >>= 8;
  int sign = (int)(r & 0x1);
  d.rabs = (r >> 1) & 0x000fffffffffffffULL;
  double x = (double)d.rabs * tab.wi(d.idx);
  d.x = sign ? -x : x;
  d.accept = d.rabs < tab.ki(d.idx);
  return d;
}
__device__ __forceinline__ ZigDraw zig_fast(uint64_t r) { return zig_fast(r, ZigTabConst{}); }
// Everything after a failed fast-path test of draw `d` (tail / wedge / full redraws), over a
// source of raw 64-bit outputs; false when the source ran dry before the draw was resolved.
__device__ __forceinline__ double u64_to_unit(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }
template <class Src>
__device__ inline bool zig_slow_from(Src &src, ZigDraw d, double &out) {
  uint64_t r;
  for (;;) {
    if (d.idx == 0) {
      for (;;) {
        if (!src.next(r)) return false;
        double xx = -AEHMC_ZIG_NOR_INV_R * log1p(-u64_to_unit(r));
        if (!src.next(r)) return false;
        double yy = -log1p(-u64_to_unit(r));
        if (yy + yy > xx * xx) {
          out = ((d.rabs >> 8) & 0x1) ? -(AEHMC_ZIG_NOR_R + xx) : AEHMC_ZIG_NOR_R + xx;
          return true;
        }
      }
    } else {
      if (!src.next(r)) return false;
      if (((c_zig_fi[d.idx - 1] - c_zig_fi[d.idx]) * u64_to_unit(r) + c_zig_fi[d.idx]) < exp(-0.5 * d.x * d.x)) {
        out = d.x;
        return true;
      }
    }
    if (!src.next(r)) return false;
    d = zig_fast(r);
    if (d.accept) {
      out = d.x;
      return true;
    }
  }
}
struct PcgSrc {  // the generator itself: never dry
  Pcg64 &g;
  __device__ __forceinline__ bool next(uint64_t &r) {
    r = pcg_next64(g);
    return true;
  }
};
__device__ inline double zig_slow(Pcg64 &rng, ZigDraw d) {
  PcgSrc src{rng};
  double z = 0.0;
  zig_slow_from(src, d, z);
  return z;
}
__device__ inline double rng_standard_normal(Pcg64 &rng) {
  ZigDraw d = zig_fast(pcg_next64(rng));
  return d.accept ? d.x : zig_slow(rng, d);
}

// ---- binomial(1, p) --------------------------------------------------------------
__device__ inline int binomial1_inversion(Pcg64 &rng, double p) {
  double q = 1.0 - p;
  double U = pcg_next_double(rng);
  // qn = exp(log(q)) is q up to a few ulp, and the loop below returns 0 when U <= qn, else 1
  // unless U - qn > p*qn/q (U within ulps of 1).  Away from those two boundaries the outcome is
  // settled without evaluating qn; inside the guard bands (probability ~1e-12) the literal
  // arithmetic decides, so the result always equals the literal one.
  const double guard = 1e-12;
  if (U < q * (1.0 - guard)) return 0;
  if (U > q * (1.0 + guard) && U < 1.0 - guard) return 1;
  double qn = exp(1 * log(q));
  // numpy: bound = min(n, n p + 10 sqrt(n p q + 1)); with n == 1 the second term is >= 10
  const long long bound = 1;
  long long X = 0;
  double px = qn;
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
  return mk128(shfl_u64(v.hi, src), shfl_u64(v.lo, src));
}

// Draws z_0..z_{n-1} exactly as n sequential Generator.normal() calls would, with the
// 64 lanes of the wave evaluating 64 consecutive stream positions at once (LCG jump-ahead).
// A rejected fast-path test (0.7 % of draws) is resolved by the scalar path reading the raw
// outputs the following lanes already hold; those lanes are then skipped as uniforms that
// were consumed, and the lanes behind them keep their draws (shifted down in the output).
// Only a rejection whose redraws run past lane 63 restarts the wave behind it.
// `store(i, z)` is called by exactly one lane per element.
struct LaneSrc {  // raw outputs of lanes idx, idx+1, ... of the current round (wave-uniform walk)
  uint64_t raw;
  int idx;
  __device__ __forceinline__ bool next(uint64_t &r) {
    if (idx > 63) return false;
    r = shfl_u64(raw, idx);
    idx++;
    return true;
  }
};
// the per-lane jump constants of a generator: A^(lane+1) and G_(lane+1) * inc (inc never changes)
struct PcgLaneJump {
  u128 Ak, GI;
};
__device__ __forceinline__ PcgLaneJump pcg_lane_jump(const Pcg64 &rng) {
  const int lane = threadIdx.x & 63;
  PcgLaneJump j;
  j.Ak = mk128(c_pcg_jump[lane][0], c_pcg_jump[lane][1]);
  j.GI = mul128(mk128(c_pcg_jump[lane][2], c_pcg_jump[lane][3]), rng.inc);
  return j;
}
template <class Store, class Tab>
__device__ inline void wave_normals(Pcg64 &rng, long long n, Store store, const Tab &tab, const PcgLaneJump &jump);
template <class Store, class Tab = ZigTabConst>
__device__ inline void wave_normals(Pcg64 &rng, long long n, Store store, const Tab &tab = Tab{}) {
  wave_normals(rng, n, store, tab, pcg_lane_jump(rng));
}
// (`jump` = pcg_lane_jump(rng), computed once by kernels that call this in a loop)
template <class Store, class Tab>
__device__ inline void wave_normals(Pcg64 &rng, long long n, Store store, const Tab &tab, const PcgLaneJump &jump) {
  const int lane = threadIdx.x & 63;
  const u128 Ak = jump.Ak, GI = jump.GI;
  long long pos = 0;  // normals delivered so far
  while (pos < n) {
    const u128 sk = muladd128(Ak, rng.state, GI);  // state after lane+1 steps
    const uint64_t raw = pcg_output(sk);
    const ZigDraw d = zig_fast(raw, tab);
    const unsigned long long fail = __ballot(!d.accept);
    const long long want = n - pos;
    if (fail == 0 && want >= 64) {  // 64 % of the rounds
      store(pos + lane, d.x);
      pos += 64;
      rng.state = shfl_u128(sk, 63);
      continue;
    }
    if (want < 64 && (fail & ((1ULL << want) - 1)) == 0) {
      // the request ends inside this round and none of the lanes it takes was rejected (round 5; 2/3 of the last rounds of
      // a D = 100 draw): lane l delivers normal pos + l, the stream stands behind lane want - 1 -- what the general
      // code below arrives at through its mask walk and shifted stores
      if (lane < want) store(pos + lane, d.x);
      rng.state = shfl_u128(sk, (int)want - 1);
      return;
    }
    // Rejections are resolved in lane order; `dropped` collects the lanes that deliver nothing
    // (consumed by a rejection's redraws, or cut off behind a restart), and every surviving
    // lane stores once at the end, shifted down by the dropped lanes below it.
    //
    // The common rejection -- a wedge test (idx != 0), 99.6 % of them -- needs ONE more uniform, which
    // is the raw output the NEXT lane already holds, and its outcome is a pure function of the two
    // raw values.  So every failing lane evaluates its own wedge test in parallel (one vector exp for
    // the whole wave instead of a wave-uniform one per rejection), and the walk below only shifts
    // bit masks: an accepted wedge drops lane f+1 (consumed as the uniform) and lane f delivers its x;
    // a rejected one drops lanes f and f+1, and the redraw IS the draw that starts at lane f+2
    // (delivered by that lane itself).  Tail draws (idx == 0) and a rejection in lane 63 (its
    // uniform lies in the next round) keep the generic scalar path.
    unsigned long long wacc = 0, tailm = 0;
    if (fail) {
      const uint64_t raw_next = (uint64_t)__shfl_down((unsigned long long)raw, 1);
      bool w = false;
      if (!d.accept && d.idx != 0)
        w = ((tab.fi(d.idx - 1) - tab.fi(d.idx)) * u64_to_unit(raw_next) + tab.fi(d.idx)) < exp(-0.5 * d.x * d.x);
      wacc = __ballot(w);
      tailm = __ballot(!d.accept && d.idx == 0);
    }
    double x = d.x;
    unsigned long long dropped = 0;
    unsigned long long wlive = 0;  // lanes that deliver through an accepted wedge: the stream stands one lane further
    int cur = 0;                // rejections are looked for at lanes >= cur
    bool serial = false;        // rng.state was advanced by the generator itself (restart)
    int ev_f = -1, ev_end = -1; // last rejection resolved on the generic path: its lane, the last lane it consumed
    for (;;) {
      const unsigned long long m = cur < 64 ? (fail & (~0ULL << cur)) : 0ULL;
      if (!m) break;
      const int f = __ffsll((long long)m) - 1;
      const int before = f - __popcll(dropped & ((1ULL << f) - 1));  // normals delivered by lanes < f
      if ((long long)before >= want) break;                           // the request ends before lane f
      if (f < 63 && !((tailm >> f) & 1ULL)) {  // wedge: masks only
        if ((wacc >> f) & 1ULL) {
          dropped |= 1ULL << (f + 1);
          wlive |= 1ULL << f;
        } else {
          dropped |= 3ULL << f;
        }
        cur = f + 2;
        continue;
      }
      ZigDraw df;
      df.x = __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(d.x), f));
      df.rabs = shfl_u64(d.rabs, f);
      df.idx = __builtin_amdgcn_readlane(d.idx, f);
      df.accept = false;
      LaneSrc src{raw, f + 1};
      double z;
      ev_f = f;
      if (zig_slow_from(src, df, z)) {  // lanes f+1 .. src.idx-1 were consumed
        const int k = src.idx - (f + 1);
        if (k) dropped |= ((1ULL << k) - 1) << (f + 1);
        ev_end = src.idx - 1;
        cur = src.idx;
      } else {  // ran past lane 63: redo this draw on the generator itself, restart behind it
        rng.state = shfl_u128(sk, f);
        z = zig_slow(rng, df);
        serial = true;
        if (f < 63) dropped |= ~0ULL << (f + 1);
        cur = 64;
      }
      if (lane == f) x = z;
    }
    const bool mine = !((dropped >> lane) & 1ULL);
    const int o = lane - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dropped >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)dropped, 0u));
    const int produced = 64 - __popcll(dropped);
    if ((long long)produced <= want) {
      if (mine) store(pos + o, x);
      pos += produced;
      if (!serial) rng.state = shfl_u128(sk, 63);
    } else {  // the request ends inside this round (never behind a restart: that lane delivers the last one)
      if (mine && o < want) store(pos + o, x);
      const unsigned long long lastm = __ballot(mine && o == want - 1);
      const int L = __ffsll((long long)lastm) - 1;
      rng.state = shfl_u128(sk, ((wlive >> L) & 1ULL) ? L + 1 : (L == ev_f ? ev_end : L));
      return;
    }
  }
}

}  // namespace aehmc
