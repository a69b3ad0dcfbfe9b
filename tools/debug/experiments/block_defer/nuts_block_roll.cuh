// Block-resident dense NUTS, several transitions per launch, chains ROLLING ON (gfx950): 64 < D <= 256.
//
// k_nuts_block_reg (nuts_block_reg.cuh) runs the transitions of a workgroup's 16 chains together: every transition lasts
// as long as the deepest of 16 trees (28 rounds at D = 200 where the mean tree has 17.3 leapfrogs).  Here a chain whose
// tree has ended records its transition, draws its next momentum and begins again while its neighbours are still in
// their trees -- the products of a round serve whatever mixture of running and beginning chains there is (row masks);
// a round in which chains begin carries one more product (p = L^-T z).  Same stage / bookkeeping arithmetic as
// k_nuts_block_reg, chain by chain: the results do not depend on the schedule, bit for bit (tests/test_gpu_block_dense.py).
// Measured (profiles/r4/INDEX.md): rounds per transition 27.2 -> 21.2, products -13 %; what of that is left after the
// compiler's handling of the larger loop body is 3-5 % at D = 200 - 256 and a loss below D ~ 190 (short products: the extra
// product of a begin round and the exposed first K-tile weigh more), so the engine uses this kernel for D >= 192 and
// launches of more than one transition (option "block_roll").
// Reference: nuts.py:56-153, trajectory.py:154-374,428-714, termination.py:85-235, proposals.py:19-174,
// integrators.py:54-73, metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "nuts_block_reg.cuh"

namespace aehmc {

// The D standard normals of a chain's next momentum (site #1: generator 0 of the chain, home in LDS at `park`) into its
// operand row.  A real call, not inlined: the draw wants ~90 registers of its own, and inside the round loop of
// k_nuts_block_reg they would come out of the chain state's 128 (spill code all over the hot path); as a callee it
// saves what it needs around itself, once per transition.
__device__ __attribute__((noinline)) void blk_draw_normals(double *park, double *xrow, long long D, int lane) {
  Pcg64 g0 = blk_gen_load(park, 0);
  wave_normals(g0, D, [=](long long i, double z) { xrow[i] = z; });
  blk_gen_store(park, 0, g0, lane);
  __threadfence_block();
}

template <int R, bool TDENSE>
__global__ __launch_bounds__(BLK_THREADS) void k_nuts_block_roll(EngineArgs a, NutsSampleArgs m) {
  extern __shared__ __attribute__((aligned(16))) double blk_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c0 = (long long)blockIdx.x * BLK_CHAINS, c = c0 + wave;
  const bool valid = c < a.C;
  const long long D = a.D;
  const int S = (int)blk_lds_stride(D);
  double *const xbuf = blk_lds, *const ybuf = blk_lds + BLK_CHAINS * S;
  double *const tb = blk_lds + 2 * BLK_CHAINS * S + wave * BLK_TB;
  double *const mus = blk_lds + 2 * BLK_CHAINS * S + BLK_CHAINS * BLK_TB;  // the dense target's mean
  // Register budget: 16 wavefronts per workgroup leave a lane 128 registers.  The chain's q, p, v and momentum sum
  // (8 R registers), its tree state (wave-uniform, but the product of fp64 VALU arithmetic: ~26 VGPRs) and the
  // product's ~45 fit; dU/dq and w do not have to be in registers (they stay in the LDS rows the products write), and
  // the four generators (32 more: 64-bit integer VALU arithmetic) have their home in LDS, a draw holding one of them
  // for its own duration.  A first version with q, p, g, v, w and the generators in registers spilled 270-450 bytes
  // per lane: the spill code alone moved 2.9 TB/s through HBM and the kernel waited on it (profiles/r4/INDEX.md);
  // now the per-leapfrog loop has no scratch access.
  double *const park = mus + S + wave * BLK_PARK;
  double *const xrow = xbuf + wave * S, *const yrow = ybuf + wave * S;
  const size_t row = (size_t)(valid ? c : 0) * D;
  const bool elem = target_is_elem(a.tkind);
  for (int k = lane; k < S; k += 64) {  // pads (and the rows of chains past C) stay zero for the whole launch
    xrow[k] = 0.0;
    yrow[k] = 0.0;
  }
  if (TDENSE)
    for (int k = threadIdx.x; k < S; k += BLK_THREADS) mus[k] = k < D ? a.mu[k] : 0.0;
#define EI(r) (lane + 64 * (r))
#define AT(ptr, r) ((ptr) + row)[EI(r)]
// The chain's row of a [C, D] array from code that runs once per transition or expansion inside the round loop: the lane
// index passes through blk_opaque first, so that the address is the array's row base (wave-uniform: scalar registers)
// plus a lane offset formed on the spot.  Left visible, the loop-invariant per-lane addresses of all ~20 arrays are
// hoisted out of the round loop as 64-bit VGPR pairs, spilled, and reloaded one by one behind an s_waitcnt that also
// waits for the previous store's acknowledgement -- ~60 dependent L2 round trips per tree start (measured: the
// beginning of a transition cost four rounds' worth of bookkeeping, and every round waited for it at the barrier).
#define ATL(ptr, r) ((ptr) + row)[lc + 64 * (r)]
  bool ok[R];
  // q, p, v and the sub-trajectory momentum sum in registers; dU/dq and w = imm dU/dq stay where the products leave
  // them, in the chain's rows of the two LDS buffers (dense target: P r lands in ybuf, imm g' in xbuf; coordinate-wise
  // target: g' is written to xbuf as the operand, imm g' lands in ybuf) -- 16 registers less at R = 4
  double q[R], p[R], v[R], pb[R];
  double *const grow = TDENSE ? yrow : xrow, *const wrow = TDENSE ? xrow : yrow;
  // The lock-step engine's work vectors (engine.hip ws_layout: one after the other, the same distance apart) are
  // addressed from ONE base and stride: two scalar register pairs instead of twenty-five -- with every pointer of
  // EngineArgs live in the round loop the scalar file overflowed into VGPR lanes, and the v_readlane traffic of getting
  // them back (1215 sites against 293) made the per-leapfrog code 25 % slower (profiles/r4/INDEX.md).
  // launch_nuts_block_roll checks the layout it relies on.  w = imm dU/dq of the four proposal slots (beside slot_q /
  // slot_p / slot_g) lives in the rows cur_w / cur_v / vhalf (which this kernel does not use otherwise) and blk_w.
  // Pointers that only the once-per-transition code uses (the state and diagnostics arrays, the per-transition records)
  // are read from a table in LDS where they are used -- as kernel arguments referenced inside the round loop each would
  // hold a scalar register pair (or a VGPR lane and a v_readlane) for the whole loop.
  __shared__ void *blk_cold[12];
  if (threadIdx.x == 0) {
    blk_cold[0] = a.q; blk_cold[1] = a.g; blk_cold[2] = a.U;
    blk_cold[3] = a.out.momentum; blk_cold[4] = a.out.acceptance_probability; blk_cold[5] = a.out.num_doublings;
    blk_cold[6] = a.out.is_turning; blk_cold[7] = a.out.is_diverging; blk_cold[8] = a.out.n_leapfrog;
    blk_cold[9] = m.samples; blk_cold[10] = m.acc_hist; blk_cold[11] = m.div_hist;
  }
#define COLD(T, i) (static_cast<T *>(blk_cold[i]))
  double *const wsb = a.cur_q;
  const size_t wss = (size_t)(a.cur_p - a.cur_q);
  const int wmd = 20 + 2 * a.max_exp;  // first vector of the dense-metric group
#define WSV(k) (wsb + (size_t)(k) * wss)
#define W_END_Q(e) WSV(3 + 3 * (e))
#define W_END_P(e) WSV(4 + 3 * (e))
#define W_END_G(e) WSV(5 + 3 * (e))
// four proposal slots (nuts_block_tree.inc): 0, 1 = the lock-step engine's two, 2 = its moving-end vectors cur_q / cur_p /
// cur_g (vectors 0, 1, 2: free here, the moving end lives in registers), 3 = psub / rbuf / zbuf (free here too)
#define W_SLOT_Q(s) WSV((s) < 2 ? 9 + 3 * (s) : ((s) == 2 ? 0 : 16))
#define W_SLOT_P(s) WSV((s) < 2 ? 10 + 3 * (s) : ((s) == 2 ? 1 : 18 + 2 * a.max_exp))
#define W_SLOT_G(s) WSV((s) < 2 ? 11 + 3 * (s) : ((s) == 2 ? 2 : 19 + 2 * a.max_exp))
#define W_PSUM WSV(15)
#define W_END_V(e) WSV(wmd + 1 + (e))
#define W_END_W(e) WSV(wmd + 4 + a.max_exp + (e))
#define W_SLOT_W(s) WSV((s) == 0 ? wmd + 3 + a.max_exp : ((s) == 1 ? wmd : ((s) == 2 ? 17 + 2 * a.max_exp : wmd + 6 + a.max_exp)))  /* cur_w, cur_v, vhalf, blk_w */
#pragma unroll
  for (int r = 0; r < R; r++) {
    ok[r] = valid && EI(r) < D;
    q[r] = p[r] = v[r] = pb[r] = 0.0;
  }
  ChainCtl ct = {};
  ct.done = 1;
  double eps = 0.0;
  // what a chain carries from one transition to the next -- its potential energy, its leapfrog total, the number of
  // transitions it has completed -- is touched once per transition: parked in LDS beside the generators
  double *const U_home = park + 16;
  long long *const nleap_home = reinterpret_cast<long long *>(park + 17), *const t_home = reinterpret_cast<long long *>(park + 18);
  if (valid) {
    const ChainRng rng0 = rng_load(a, c);
#pragma unroll
    for (int k = 0; k < 4; k++) blk_gen_store(park, k, rng0.g[k], lane);
    eps = a.eps_c ? a.eps_c[c] : a.eps;
    if (lane == 0) {
      *U_home = a.U[c];
      *nleap_home = 0;
      *t_home = 0;
    }
  }
  BlkTimer tm;
  int roles = 0, pend = 0;  // (nuts_block_tree.inc: which proposal slot plays which role; what deferred() has to do)

#define BT_LC (blk_opaque(lane) & 63)
#define BTA(ptr, r) ATL(ptr, r)
#define BT_SLOT_Q(s) W_SLOT_Q(s)
#define BT_SLOT_P(s) W_SLOT_P(s)
#define BT_SLOT_G(s) W_SLOT_G(s)
#define BT_END_Q(e) W_END_Q(e)
#define BT_END_P(e) W_END_P(e)
#define BT_END_G(e) W_END_G(e)
#define BT_END_V(e) W_END_V(e)
#define BT_END_W(e) W_END_W(e)
#define BT_PSUM W_PSUM
#define BT_TAKE_W(slot, r, w) ATL(W_SLOT_W(slot), r) = (w)  /* w = imm dU/dq travels with the proposal: the next transition starts from it */
#define BT_HAS_W 1
#define BT_OUT_Q COLD(double, 0)
#define BT_OUT_G COLD(double, 1)
#define BT_OUT_U COLD(double, 2)
#define BT_OUT_MOM COLD(double, 3)
#define BT_OUT_ACC COLD(double, 4)
#define BT_OUT_NDOUBL COLD(int64_t, 5)
#define BT_OUT_TURN COLD(int32_t, 6)
#define BT_OUT_DIV COLD(int32_t, 7)
#define BT_OUT_NLEAP COLD(int64_t, 8)
#include "nuts_block_tree.inc"
#undef BT_LC
#undef BTA
#undef BT_SLOT_Q
#undef BT_SLOT_P
#undef BT_SLOT_G
#undef BT_END_Q
#undef BT_END_P
#undef BT_END_G
#undef BT_END_V
#undef BT_END_W
#undef BT_PSUM
#undef BT_TAKE_W
#undef BT_HAS_W
#undef BT_OUT_Q
#undef BT_OUT_G
#undef BT_OUT_U
#undef BT_OUT_MOM
#undef BT_OUT_ACC
#undef BT_OUT_NDOUBL
#undef BT_OUT_TURN
#undef BT_OUT_DIV
#undef BT_OUT_NLEAP

  // ---- scheduling: chains roll on ----------------------------------------------------------------------------------
  // A chain whose tree has ended does not wait for the deepest tree of its workgroup: it records the transition, in the
  // next round -- while the running chains do their bookkeeping, so that nobody waits for it at the barrier -- draws
  // the next momentum's normals into its operand row (protected from the products by their row masks) and WAITS; once
  // `roll` chains wait (or nothing else runs) the round carries one more product -- p = L^-T z for the waiting rows --
  // and the products of the round form v = imm p beside the running chains' rows; w = imm dU/dq of the accepted state
  // is the one the leapfrog that produced it computed (kept with the proposal: slot_w), so the chain is back in the
  // next round.  With the trees of a warmed-up sampler (12 or 21 leapfrogs at D = 200, rarely 27 / 35) a launch of T
  // transitions costs ~T x (mean + a begin round) rounds instead of T x the deepest tree of 16 (profiles/r4/INDEX.md:
  // the estimate from recorded tree lengths and the measurement).  Only the FIRST transition of a launch has to form w
  // (one more product, all chains together).  Per-chain arithmetic is untouched: same bits whatever the schedule.
  // (PH_END: the tree has ended; its deferred tree code and the transition's records run under the next round's products)
  constexpr int PH_RUN = 0, PH_WAIT = 1, PH_DONE = 2, PH_DRAW = 3, PH_END = 4;
  __shared__ int blk_status[BLK_CHAINS];
  const int roll = m.roll > 0 ? m.roll : (TDENSE ? 3 : 4);
  int phase = valid ? PH_WAIT : PH_DONE;
  // the normals of the next momentum (site #1) into the chain's operand row
  auto draw_z = [&]() __attribute__((always_inline)) { blk_draw_normals(park, xrow, D, lane); };
  // nuts_init_chain<true>: q, p, v in registers, dU/dq and w in their rows
  auto init_tree = [&]() __attribute__((always_inline)) {
    const int lc = blk_opaque(lane) & 63;  // (addresses formed here, not hoisted out of the round loop: see ATL)
    double kd = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      pb[r] = 0.0;
      if (ok[r]) {
        const double gr = grow[EI(r)], wr = wrow[EI(r)];
        kd += v[r] * p[r];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          ATL(W_END_Q(e), r) = q[r];
          ATL(W_END_P(e), r) = p[r];
          ATL(W_END_G(e), r) = gr;
          ATL(W_END_V(e), r) = v[r];
          ATL(W_END_W(e), r) = wr;
        }
        ATL(W_SLOT_Q(0), r) = q[r];
        ATL(W_SLOT_P(0), r) = p[r];
        ATL(W_SLOT_G(0), r) = gr;
        ATL(W_SLOT_W(0), r) = wr;
        ATL(W_PSUM, r) = p[r];
      }
    }
    kd = wave_sum(kd);
    const double U = *U_home;
    ct.H0 = U + 0.5 * kd;
    ct.U_cur = U;
    init_parked(U, ct.H0);
    ct.nleap = 0;
    ct.j = 0;
    ct.tmin = ct.tmax = 0;
    ct.done = ct.phantom = 0;
    ct.ndoubl = ct.out_div = ct.out_turn = 0;
    ct.dir = blk_bernoulli(park, 1, 0.5, lane);  // trajectory.py:516
    ct.step = 0;
  };
  // per-transition records (the outputs themselves were written when the transition ended), then WAIT or DONE
  auto end_transition = [&]() __attribute__((always_inline)) {
    const int lc = blk_opaque(lane) & 63;  // (addresses formed here, not hoisted out of the round loop: see ATL)
    const long long t_cur = *t_home;
    if (lane == 0) {
      *U_home = slot_U[ct.prop_slot];
      *nleap_home += ct.nleap;
      *t_home = t_cur + 1;
      if (COLD(double, 10)) COLD(double, 10)[(size_t)t_cur * a.C + c] = dstate[6];
      if (COLD(int, 11)) COLD(int, 11)[(size_t)t_cur * a.C + c] = ct.out_div;
    }
    if (COLD(double, 9)) {
      double *dst = COLD(double, 9) + ((size_t)t_cur * a.C + c) * D;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (ok[r]) dst[lc + 64 * r] = ATL(COLD(double, 0), r);
    }
    phase = t_cur + 1 < m.T ? PH_DRAW : PH_DONE;
  };

  if (phase == PH_WAIT) draw_z();
  tm.tick(7);
  bool first = true;  // the launch's first round: every chain begins, and w = imm dU/dq has to be formed
  // ---- one round: a leapfrog of every running chain (first stages | P r | imm g' | last stage + bookkeeping), the
  //      beginning of the next transition of the waiting ones when enough of them wait ----
  for (;;) {
    if (phase == PH_RUN) stage12();
    tm.tick(5);
    if (lane == 0) blk_status[wave] = phase;
    blk_barrier_lds();
    unsigned runmask = 0, waitmask = 0, drawmask = 0;
#pragma unroll
    for (int k = 0; k < BLK_CHAINS; k++) {
      const int sk = blk_status[k];
      runmask |= (sk == PH_RUN ? 1u : 0u) << k;
      waitmask |= (sk == PH_WAIT ? 1u : 0u) << k;
      drawmask |= ((sk == PH_DRAW || sk == PH_END) ? 1u : 0u) << k;
    }
    runmask = __builtin_amdgcn_readfirstlane(runmask);
    waitmask = __builtin_amdgcn_readfirstlane(waitmask);
    drawmask = __builtin_amdgcn_readfirstlane(drawmask);
    tm.tick(6);  // vote (waits for the slowest chain's bookkeeping)
    if (!(runmask | waitmask | drawmask)) break;
    const int nrun = __popc(runmask), nwait = __popc(waitmask);
    const int quarter = (nrun + nwait) / 4;  // (towards the end of the launch few chains are left to wait for)
    const int need = roll >= BLK_CHAINS ? BLK_CHAINS + 1 : (roll < quarter ? roll : (quarter > 1 ? quarter : 1));
    // rows that begin a transition in this round (nothing running: all that wait, once the last normals are drawn)
    const unsigned bmask = (nwait >= need || (nrun == 0 && drawmask == 0)) ? waitmask : 0u;
    const bool beginning = (bmask >> wave) & 1u;
    // the tree code that book() deferred (nuts_block_tree.inc) and, behind a tree's end, the transition's records: under
    // the products -- in the phase of the product the wavefront has no column block of, else before its block of the
    // first or of the second product by turns (one call site: the product phases are a loop)
    auto window = [&]() __attribute__((always_inline)) {
      if (pend) deferred();
      if (phase == PH_END) {
        end_transition();
        tm.tick(7);
      }
    };
    const bool has_work = pend != 0 || phase == PH_END;
    if (!(runmask | bmask)) {
      // (the workgroup's last trees have just ended: their chains draw below, the round has no products)
      if (has_work) window();
      blk_barrier_lds();  // (blk_status is rewritten at the head of the next round)
    } else if (TDENSE) {
      const int when = !blk_has_tile(D, wave, false) ? 0 : (!blk_has_tile(D, wave, true) ? 1 : (wave >> 2) & 1);
#pragma nounroll
      for (int ph = 0; ph < 2; ph++) {
        if (ph == when && has_work) window();
        if (ph == 0) {
          if (bmask) blk_gemm_lds(xbuf, ybuf, S, a.sqrt_mass, D, wave, lane, tb, bmask);  // p = L^-T z (metrics.py:66-67)
          if (runmask) blk_gemm_lds(xbuf, ybuf, S, m.prec, D, wave, lane, tb, runmask);   // dU/dq' = P r
        } else {
          blk_gemm_lds<true>(ybuf, xbuf, S, a.imm, D, wave, lane, tb, runmask | bmask);    // w' = imm dU/dq' | v = imm p
        }
        tm.tick(2);
        blk_barrier_lds();
        tm.tick(3);
        if (ph == 0 && phase == PH_RUN) stash_g();  // (dU/dq' into the candidate slot: acknowledged under the second product)
        if (__builtin_expect(beginning, 0)) {
          if (ph == 0) {
#pragma unroll
            for (int r = 0; r < R; r++) p[r] = ok[r] ? yrow[EI(r)] : 0.0;
          } else {
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = ok[r] ? xrow[EI(r)] : 0.0;
          }
        }
      }
    } else {
      if (has_work) window();
      if (bmask) blk_gemm_lds(xbuf, ybuf, S, a.sqrt_mass, D, wave, lane, tb, bmask);  // p = L^-T z
      if (runmask) blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb, runmask);    // w' = imm dU/dq'
      tm.tick(2);
      blk_barrier_lds();
      tm.tick(3);
      if (__builtin_expect(beginning, 0)) {  // p becomes the operand row of v = imm p
#pragma unroll
        for (int r = 0; r < R; r++) {
          p[r] = ok[r] ? yrow[EI(r)] : 0.0;
          if (ok[r]) xrow[EI(r)] = p[r];
        }
      }
    }
    if (phase == PH_RUN) {  // (one call site: the bookkeeping is inlined once)
      book();
      if (ct.done) phase = PH_END;
      tm.tick(4);
    }
    if (!TDENSE && __builtin_expect(bmask != 0, 0)) {  // (the running chains' bookkeeping above filled the wait)
      blk_barrier_lds();
      blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb, bmask);  // v = imm p
      tm.tick(2);
      blk_barrier_lds();
      tm.tick(3);
      if (beginning) {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = ok[r] ? yrow[EI(r)] : 0.0;
      }
    }
    if (__builtin_expect(beginning, 0)) {  // the state the transition starts from: position, dU/dq -> its row, w -> its row
      const int s_acc = ct.prop_slot;
      const int lc = blk_opaque(lane) & 63;
#pragma unroll
      for (int r = 0; r < R; r++) {
        q[r] = ok[r] ? ATL(COLD(double, 0), r) : 0.0;
        if (ok[r]) {
          grow[EI(r)] = ATL(COLD(double, 1), r);
          if (!first) wrow[EI(r)] = ATL(W_SLOT_W(s_acc), r);
        }
      }
    }
    if (__builtin_expect(first, 0)) {  // w = imm dU/dq (grow's buffer -> wrow's buffer)
      blk_barrier_lds();
      if (TDENSE) blk_gemm_lds(ybuf, xbuf, S, a.imm, D, wave, lane, tb, bmask);
      else blk_gemm_lds(xbuf, ybuf, S, a.imm, D, wave, lane, tb, bmask);
      blk_barrier_lds();
      tm.tick(2);
    }
    if (__builtin_expect(phase == PH_DRAW, 0)) {
      draw_z();
      phase = PH_WAIT;
      tm.tick(7);
    } else if (__builtin_expect(beginning, 0)) {
      init_tree();
      phase = PH_RUN;
      tm.tick(7);
    }
    first = false;
  }
  if (valid) {
    ChainRng rng1;
#pragma unroll
    for (int k = 0; k < 4; k++) rng1.g[k] = blk_gen_load(park, k);
    rng_store(a, c, lane, rng1, 0, 3);
    if (lane == 0 && m.nleap_total) m.nleap_total[c] = *nleap_home;
#ifdef AEHMC_WIDE_TIMING
    if (lane == 0)
      for (int k = 0; k < 16; k++) a.linreg_part[c * 16 + k] = (double)tm.acc[k];
#endif
  }
  (void)elem;
#undef EI
#undef AT
#undef ATL
#undef COLD
#undef WSV
#undef W_END_Q
#undef W_END_P
#undef W_END_G
#undef W_SLOT_Q
#undef W_SLOT_P
#undef W_SLOT_G
#undef W_PSUM
#undef W_END_V
#undef W_END_W
#undef W_SLOT_W
}

// k_nuts_block_roll addresses the work vectors as cur_q + k * (cur_p - cur_q) (engine.hip ws_layout)
inline bool blk_roll_layout_ok(const EngineArgs &a) {
  const ptrdiff_t s = a.cur_p - a.cur_q;
  const ptrdiff_t E = a.max_exp, md = 20 + 2 * E;
  auto at = [&](const double *p, ptrdiff_t k) { return p == a.cur_q + k * s; };
  bool ok = s > 0 && at(a.cur_g, 2) && at(a.psum, 15) && at(a.psub, 16) && at(a.ckp, 17) && at(a.vhalf, 17 + 2 * E) &&
            at(a.rbuf, 18 + 2 * E) && at(a.zbuf, 19 + 2 * E) && at(a.cur_v, md) && at(a.ckv, md + 3) &&
            at(a.cur_w, md + 3 + E) && at(a.blk_w, md + 6 + E);
  for (int e = 0; e < 2; e++)
    ok = ok && at(a.end_q[e], 3 + 3 * e) && at(a.end_p[e], 4 + 3 * e) && at(a.end_g[e], 5 + 3 * e) &&
         at(a.slot_q[e], 9 + 3 * e) && at(a.slot_p[e], 10 + 3 * e) && at(a.slot_g[e], 11 + 3 * e) &&
         at(a.end_v[e], md + 1 + e) && at(a.end_w[e], md + 4 + E + e);
  return ok;
}
template <int R>
inline hipError_t launch_nuts_block_roll_r(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) {
  const size_t dyn = blk_reg_lds_bytes(a.D);
  const dim3 grid((unsigned)((a.C + BLK_CHAINS - 1) / BLK_CHAINS)), block(BLK_THREADS);
#define AEHMC_BLK(TDV)                                                                                       \
  do {                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_block_roll<R, TDV>),           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                \
    if (e != hipSuccess) return e;                                                                           \
    hipLaunchKernelGGL((k_nuts_block_roll<R, TDV>), grid, block, dyn, st, a, m);                             \
  } while (0)
  if (a.tkind == AEHMC_T_DENSE_MVN) AEHMC_BLK(true);
  else AEHMC_BLK(false);
#undef AEHMC_BLK
  return hipGetLastError();
}
// does a call of m.T transitions at this D go to the rolling kernel?  roll: option "block_roll".  Measured, 4096 chains,
// sample(10), rolling against transition by transition: D = 130 0.448 / 0.404 ms, 150 0.482 / 0.436, 180 0.589 / 0.572,
// 200 0.692 / 0.726, 256 0.973 / 1.001
constexpr long long BLK_ROLL_MIN_D = 192;
inline bool block_roll_wanted(long long D, long long T, int roll) {
  if (T <= 1 || roll >= BLK_CHAINS || !block_reg_supported(D)) return false;
  return roll > 0 || D >= BLK_ROLL_MIN_D;
}
inline hipError_t launch_nuts_block_roll(EngineArgs a, NutsSampleArgs m, double *bp, hipStream_t st) {
  if (!blk_roll_layout_ok(a)) return hipErrorInvalidValue;
  BlkMats mats;
  if (hipError_t e = blk_pack_matrices(a, m.prec, bp, mats, st)) return e;
  a.imm = mats.imm; a.sqrt_mass = mats.sqrt_mass; m.prec = mats.prec;
  return a.D <= 128 ? launch_nuts_block_roll_r<2>(a, m, st) : launch_nuts_block_roll_r<4>(a, m, st);
}

}  // namespace aehmc
