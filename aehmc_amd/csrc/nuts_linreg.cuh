// NUTS on the linear-regression target (examples/LinearRegression.ipynb:126-166, q = [w, log n],
// D = 2): any number of consecutive nuts.new_kernel(...)(state, eps, imm) transitions of every
// chain in ONE launch.
//
// The target's gradient is a reduction over the N data rows, so a 512-thread workgroup owns FOUR
// chains: per leapfrog all eight wavefronts sweep (X, y) once from L2 for the four chains together
// (linreg_rows.cuh: direct 16-byte loads, double-buffered in registers, three fused multiply-adds
// per row and chain; the first 10176 rows stay in the CU's LDS for the whole launch and never
// touch the vector-memory path again), and wavefronts 0-3 each keep the tree of one chain.
// D = 2, so the whole transition state of a chain lives in registers of its wavefront -- element
// e of every vector in lane e, the U-turn checkpoints (termination.py:12-16) of level i in lanes
// 2i, 2i+1 of two more registers; the tree touches no memory at all.
//
// Chains are independent, so nothing is synchronised at transition boundaries: a chain whose tree
// has ended draws its next momentum and goes on in the very next sweep while its neighbours are
// still inside their trees (the sweep serves four chains whatever transition each is in).  With
// the short, unequal trees of a warmed-up sampler (2 or 5 leapfrogs, rarely more) a launch of T
// transitions costs about T x the MEAN tree length instead of T x the longest tree on the GPU.
// With m.adapt the launch is a whole window-adaptation warm-up (window_adaptation.py:17-116,
// diagonal mass matrix): after each of its transitions a chain applies its own dual-averaging /
// Welford / window-end update (the arithmetic of k_adapt_update) and goes on with the new
// parameters -- early warm-up trees range from 1 to 1023 leapfrogs, which is where not waiting
// for the slowest chain pays most.
//
// Diagonal / scalar metric (shared or per chain); DM instantiation (round 3): a dense 2 x 2 inverse mass matrix,
// shared or one per chain (what is_mass_matrix_full adaptation of this example hands back) -- row `lane` of the
// matrix and of L^-T in two registers, velocities formed literally (metrics.py:71: M[i][0] p0 + M[i][1] p1, the
// order of the per-chain products of the lock-step path); with m.adapt the chain adapts its full 2 x 2 matrix in
// the launch: the arithmetic of k_adapt_update's full branch and of wave_chol_inv_t written out for D = 2.
// Arithmetic and its order are those of nuts_book
// (engine.cuh) / k_nuts_resident; reference: nuts.py:56-153, trajectory.py:154-374,428-714,
// termination.py:85-235, proposals.py:19-174, integrators.py:54-73, metrics.py:44-104.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.cuh"
#include "linreg_rows.cuh"

namespace aehmc {

#ifdef AEHMC_WIDE_TIMING  // developer build (make timing): cycles per phase of every chain wave -> a.ckp[c][8]
#define LRN_TICK(k)                                                  \
  do {                                                               \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();  \
    tacc[k] += now_ - tlast;                                         \
    tlast = now_;                                                    \
  } while (0)
#else
#define LRN_TICK(k) do { } while (0)
#endif

constexpr int NUTS_LINREG_MAX_EXP = 32;  // checkpoint levels that fit the lanes of a wavefront
constexpr long long NUTS_LINREG_LDS_ROWS = 10176;  // rows kept in LDS (2 x 8 B each, next to ~1 KB static)

template <bool DM>
__global__ __launch_bounds__(LR_BLOCK) void k_nuts_linreg(EngineArgs a, NutsSampleArgs m) {
  __shared__ double lr_w[4], lr_part[LR_WAVES][8];
  __shared__ int lr_fin[4];
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  const long long NL = a.N <= NUTS_LINREG_LDS_ROWS ? a.N : NUTS_LINREG_LDS_ROWS;  // rows [0, NL) live in LDS
  for (long long i = threadIdx.x; i < NL; i += LR_BLOCK) {
    dyn_lds[i] = a.X[i];
    dyn_lds[NL + i] = a.y[i];
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long c = wave < 4 ? (long long)blockIdx.x * 4 + wave : a.C;
  bool fin = c >= a.C;  // no chain (row-serving wavefront), or all T transitions done
  const int e = lane & 1;
  const bool el = lane < 2;  // this lane holds element `lane` of the chain's vectors
  // sum over the two elements (other lanes hold nothing): wave-uniform, lane order 0, 1
  auto sum2 = [](double x) { return read_lane_f64(x, 0) + read_lane_f64(x, 1); };

  double q = 0.0, g = 0.0, p = 0.0, pb = 0.0;  // moving end + sub-trajectory momentum sum
  double sq = 0.0, sg = 0.0, U_state = 0.0;    // the chain's state between transitions
  double end_q[2], end_p[2], end_g[2], slot_q[2], slot_p[2], slot_g[2], psum = 0.0;
  double ckp = 0.0, cks = 0.0;                 // checkpoints: level i, element e in lane 2 i + e
  double imr = 1.0, smr = 0.0, eps = 0.0;
  double im0 = 0.0, im1 = 0.0, sm0 = 0.0, sm1 = 0.0;  // DM: row `lane` of the inverse mass matrix and of L^-T
  // velocity element `lane` of the momentum whose elements sit in lanes 0, 1 of pv (metrics.py:71)
  auto vel = [&](double pv) -> double {
    if (!DM) return imr * pv;
    const double p0 = read_lane_f64(pv, 0), p1 = read_lane_f64(pv, 1);
    return im0 * p0 + im1 * p1;
  };
  ChainRng rng = {};
  ChainCtl ct = {};
  long long t_idx = 0, nleap_sum = 0;
  DualAvg da = {1, 0.0, 0.0, 0.0, 0.0};  // warm-up state of the chain (m.adapt)
  long long wc_n = 0;
  double wmean = 0.0, wm2 = 0.0;
  double wm2_0 = 0.0, wm2_1 = 0.0;  // DM: row `lane` of the 2 x 2 Welford sum
#pragma unroll
  for (int s = 0; s < 2; s++) end_q[s] = end_p[s] = end_g[s] = slot_q[s] = slot_p[s] = slot_g[s] = 0.0;

  // ---- nuts.py:113-125: momentum (site #1, metrics.py:65-68), initial energy, fresh tree --------
  auto begin_transition = [&]() {
    const double z0 = rng_standard_normal(rng.g[0]), z1 = rng_standard_normal(rng.g[0]);
    p = el ? (DM ? sm0 * z0 + sm1 * z1 : smr * (lane == 0 ? z0 : z1)) : 0.0;  // metrics.py:66-67: L^-T z
    q = sq;
    g = sg;
    pb = 0.0;
    const double v_init = vel(p);
    const double kd = sum2(el ? v_init * p : 0.0);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      end_q[s] = q;
      end_p[s] = p;
      end_g[s] = g;
    }
    slot_q[0] = q;
    slot_p[0] = p;
    slot_g[0] = g;
    psum = p;
    const double U = U_state;
    ct.H0 = U + 0.5 * kd;
    ct.prop_E = ct.H0;
    ct.prop_w = 0.0;
    ct.prop_slpa = -INFINITY;
    ct.sub_E = ct.sub_w = ct.sub_slpa = 0.0;
    ct.U_cur = ct.U_end[0] = ct.U_end[1] = ct.U_slot[0] = ct.U_slot[1] = U;
    ct.acc_prob = 0.0;
    ct.nleap = 0;
    ct.j = 0;
    ct.length = 0;
    ct.tmin = ct.tmax = 0;
    ct.done = ct.phantom = 0;
    ct.prop_slot = 0;
    ct.ndoubl = ct.out_div = ct.out_turn = 0;
    ct.dir = rng_bernoulli(rng.g[1], 0.5);  // trajectory.py:516
    ct.step = 0;
  };
  if (!fin) {
    const size_t imo = (size_t)c * a.imm_cs;
    if (el) {
      sq = a.q[c * 2 + lane];
      sg = a.g[c * 2 + lane];
      if (DM) {  // [2, 2] shared, or [C, 2, 2]
        const size_t mo = a.imm_cs ? (size_t)c * 4 : 0;
        im0 = a.imm[mo + 2 * lane];
        im1 = a.imm[mo + 2 * lane + 1];
        sm0 = a.sqrt_mass[mo + 2 * lane];
        sm1 = a.sqrt_mass[mo + 2 * lane + 1];
      } else {
        imr = a.imm[imo + (a.met_ndim == 0 ? 0 : lane)];
        smr = a.sqrt_mass[imo + (a.met_ndim == 0 ? 0 : lane)];
      }
    }
    U_state = a.U[c];
    eps = a.eps_c ? a.eps_c[c] : a.eps;
    if (m.adapt) {
      da.step = m.ad.da_step[c];
      da.x = m.ad.da_x[c];
      da.x_avg = m.ad.da_x_avg[c];
      da.g_avg = m.ad.da_g_avg[c];
      da.mu = m.ad.da_mu[c];
      wc_n = m.ad.wc_n[c];
      if (el) {
        wmean = m.ad.wc_mean[c * 2 + lane];
        if (DM) {
          wm2_0 = m.ad.wc_m2[c * 4 + 2 * lane];
          wm2_1 = m.ad.wc_m2[c * 4 + 2 * lane + 1];
        } else {
          wm2 = m.ad.wc_m2[c * 2 + lane];
        }
      }
    }
    rng = rng_load(a, c);
    begin_transition();
  }

#ifdef AEHMC_WIDE_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    // ---- one leapfrog of the moving end (integrators.py:54-73), first half ------------------
    const double step_size = (ct.dir ? 1.0 : -1.0) * eps;
    const double b = 0.5 * step_size, aa = 1 * step_size;
    if (DM) {
      const double pp = p - b * g;
      const double vh = vel(pp);  // (both elements of p_half are needed: outside the lane mask)
      if (!fin && el) {
        q = q + aa * vh;
        p = pp;
      }
    } else if (!fin && el) {
      const double pp = p - b * g;
      q = q + aa * (imr * pp);
      p = pp;
    }
    if (lane == 0 && wave < 4) {
      lr_w[wave] = q;
      lr_fin[wave] = fin;
    }
    // the row-independent part of U and dU/dq (exp, log) ahead of the sweep: off the critical path
    double ww = 0.0, ell = 0.0, n = 1.0, n2 = 1.0, lp_wn = 0.0;
    if (!fin) {
      ww = read_lane_f64(q, 0);
      ell = read_lane_f64(q, 1);
      n = exp(ell);
      n2 = n * n;
      lp_wn = (-0.5 * ww * ww - AEHMC_LOG_SQRT_2PI) + (log(n) - n + ell);  // lp_w + lp_n of k_linreg_finish
    }
    __syncthreads();
    LRN_TICK(0);  // first half, publish, barrier
    if (lr_fin[0] & lr_fin[1] & lr_fin[2] & lr_fin[3]) break;
    // sum(x r) and sum(r^2), r = y - x w, over all rows for the four chains of the workgroup
    {
      double w4[4], sxr[4], srr[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        w4[k] = lr_w[k];
        sxr[k] = srr[k] = 0.0;
      }
      // (the LDS rows are added while the first streamed block is on its way)
      lr_rows_direct(a.X + NL, a.y + NL, a.N - NL, wave, lane, w4, sxr, srr, [&] {
        lr_rows_lds<4, LR_BLOCK>(dyn_lds, dyn_lds + NL, NL, (int)threadIdx.x, w4, sxr, srr);
      });
      const double v[8] = {sxr[0], sxr[1], sxr[2], sxr[3], srr[0], srr[1], srr[2], srr[3]};
      const double tot = wave_sum8(v, lane);  // lane l: wave total of v[l & 7]
      if (lane < 8) lr_part[wave][lane] = tot;
    }
    LRN_TICK(1);  // row sweep
    __syncthreads();
    LRN_TICK(2);  // barrier
    if (fin) continue;
    double s_xr = lr_part[0][wave], s_rr = lr_part[0][4 + wave];
#pragma unroll
    for (int w = 1; w < LR_WAVES; w++) {
      s_xr += lr_part[w][wave];
      s_rr += lr_part[w][4 + wave];
    }
    double kd;
    {  // U and dU/dq as k_linreg_finish, second half of the leapfrog
      const double N = (double)a.N;
      const double lp_y = -0.5 * (s_rr / n2) - N * AEHMC_LOG_SQRT_2PI - N * ell;
      const double gg = lane == 0 ? -(-ww + s_xr / n2) : -(2.0 - n - N + s_rr / n2);
      ct.U_cur = -(lp_wn + lp_y);
      double kl = 0.0;
      if (el) {
        const double pp = p - b * gg;
        g = gg;
        p = pp;
        if (!DM) kl = (imr * pp) * pp;
      }
      if (DM) {
        const double vn = vel(p);
        kl = el ? vn * p : 0.0;
      }
      kd = sum2(kl);
    }
    LRN_TICK(3);  // target finish, second half

    // ---- dynamic_integration body (trajectory.py:195-305) -----------------------------------
    const int step = ct.step;
    if (!ct.phantom) ct.nleap += 1;
    const TreeIdx ti = tree_step_indices(step, ct.tmin, ct.tmax);  // termination.py:109-113 (stale at step 0), 192-235
    const int tmin = ti.tmin, tmax = ti.tmax;
    if (el) pb = (step == 0) ? p : pb + p;
    if ((step & 1) == 0) {  // termination.py:115-131
      const double pe = __shfl(p, e), pbe = __shfl(pb, e);
      if ((lane >> 1) == tmax) {
        ckp = pe;
        cks = pbe;
      }
    }
    ct.tmin = tmin;
    ct.tmax = tmax;
    // proposals.py:19-62, progressive_uniform_sampling proposals.py:72-102 (+ :141-144) -- nuts_tree.cuh
    const TreePoint np = tree_new_point(ct.H0, ct.U_cur, kd, a.thr);
    const bool div = np.div;
    bool term = false;
    const bool take = tree_sample_step<true>(ct, step, np, lane, [&](double pr) { return rng_bernoulli(rng.g[2], pr); });
    if (step >= 1) {
      if (tmax >= tmin) {  // termination.py:133-187
        int idx = tmax;
        bool crit = false;
        for (;;) {
          const double pl = __shfl(ckp, 2 * idx + e), ks = __shfl(cks, 2 * idx + e);
          double dl = 0.0, dr = 0.0;
          const double vl_d = DM ? vel(pl) : 0.0, vr_d = DM ? vel(p) : 0.0;
          if (el) {
            const double pr = p;
            const double vl = DM ? vl_d : imr * pl, vr = DM ? vr_d : imr * pr;
            const double sub = pb - ks + pl;
            const double rho = sub - (pr + pl) / 2;
            dl = vl * rho;
            dr = vr * rho;
          }
          const double d_l = sum2(dl), d_r = sum2(dr);
          crit = (d_l <= 0) | (d_r <= 0);
          const bool reached = (idx - 1) < tmin;
          idx -= 1;
          if (crit || reached) break;
        }
        term = crit;
      }
    }
    if (take) {  // sub-trajectory proposal <- moving end (copy on accept)
      const int s = ct.prop_slot ^ 1;
      if (s) {
        slot_q[1] = q;
        slot_p[1] = p;
        slot_g[1] = g;
      } else {
        slot_q[0] = q;
        slot_p[0] = p;
        slot_g[0] = g;
      }
      put2(ct.U_slot, s, ct.U_cur);
    }

    // ---- sub-trajectory / expansion control (trajectory.py:336, 537-608) --------------------
    const TreeControl tc = tree_step_control(ct, step, div, term);
    const bool fin_div = tc.fin_div, fin_term = tc.fin_term;
    if (tc.finalize) {
      const int dir = ct.dir, oth = 1 - dir;
      double dl = 0.0, dr = 0.0;
      {
        const double pc = p, po = pick2(end_p, oth);
        const double vc = vel(pc), vo = vel(po);
        const double s = psum + pb;
        psum = s;
        const double pl = dir ? po : pc, pr = dir ? pc : po;
        const double vl = dir ? vo : vc, vr = dir ? vc : vo;
        const double rho = s - (pr + pl) / 2;
        if (el) {
          dl = vl * rho;
          dr = vr * rho;
        }
        if (dir) {
          end_q[1] = q;
          end_p[1] = pc;
          end_g[1] = g;
        } else {
          end_q[0] = q;
          end_p[0] = pc;
          end_g[0] = g;
        }
      }
      const double d_l = sum2(dl), d_r = sum2(dr);
      const bool turning = (d_l <= 0) | (d_r <= 0);
      put2(ct.U_end, dir, ct.U_cur);
      if (tree_merge_expansion<true>(ct, fin_div, fin_term, lane, [&](double pr) { return rng_bernoulli(rng.g[3], pr); }))
        ct.prop_slot ^= 1;
      const bool end_transition = tree_expansion_outcome(ct, fin_div, fin_term, turning, a.max_exp);
      if (end_transition) {
        if (step == 0 && fin_div) {  // trajectory.py:336: the scan still runs (phantom)
          ct.phantom = 1;
          ct.step = 1;
        } else {
          ct.done = 1;
        }
      } else {
        ct.j += 1;
        const int go_right = rng_bernoulli(rng.g[1], 0.5);
        ct.dir = go_right;
        ct.step = 0;
        if (go_right != dir) {  // continue from the other end
          q = pick2(end_q, go_right);
          p = pick2(end_p, go_right);
          g = pick2(end_g, go_right);
          ct.U_cur = pick2(ct.U_end, go_right);
        }
      }
    }

    LRN_TICK(4);  // tree
    // ---- the transition is over (a phantom scan cannot change its outputs): hand the proposal
    // to the chain's state, record it, start the next transition or leave --------------------
    if (ct.done) {
      const int s = ct.prop_slot;
      sq = pick2(slot_q, s);
      sg = pick2(slot_g, s);
      const double sp = pick2(slot_p, s);
      U_state = pick2(ct.U_slot, s);
      nleap_sum += ct.nleap;
      if (m.samples && el) m.samples[((size_t)t_idx * a.C + c) * 2 + lane] = sq;
      if (lane == 0) {
        if (m.acc_hist) m.acc_hist[(size_t)t_idx * a.C + c] = ct.acc_prob;
        if (m.div_hist) m.div_hist[(size_t)t_idx * a.C + c] = ct.out_div;
      }
      if (m.adapt) {  // the update of k_adapt_update, same order: step size, Welford, window end, last
        const int stage = m.stage[t_idx], wend = m.window_end[t_idx];
        double step_size = adapt_da_update(da, m.target, ct.acc_prob, m.gamma, m.t0, m.kappa);
        if (DM) {
          // full 2 x 2 covariance: k_adapt_update's full branch (algorithms.py:187-197 with np.outer,
          // mass_matrix.py:83-118) and wave_chol_inv_t (metrics.py:56-58) written out for D = 2, row `lane` per lane
          if (stage != 0) {
            wc_n += 1;
            double delta = 0.0, ud = 0.0;
            if (el) {
              delta = sq - wmean;
              wmean = wmean + delta / (double)wc_n;
              ud = sq - wmean;
            }
            const double d0 = read_lane_f64(delta, 0), d1 = read_lane_f64(delta, 1);
            wm2_0 = wm2_0 + ud * d0;
            wm2_1 = wm2_1 + ud * d1;
          }
          if (wend) {
            const double nn = (double)wc_n;
            double c0 = wm2_0 / (double)(wc_n - 1), c1 = wm2_1 / (double)(wc_n - 1);
            im0 = (nn / (nn + 5)) * c0;
            im1 = (nn / (nn + 5)) * c1;
            if (lane == 0) im0 = im0 + 1e-3 * (5 / (nn + 5));  // shrinkage * eye
            if (lane == 1) im1 = im1 + 1e-3 * (5 / (nn + 5));
            wm2_0 = wm2_1 = 0.0;
            wmean = 0.0;
            // L = chol(imm) from its lower triangle, S = L^-T (the column-by-column forward substitution for D = 2)
            const double a00 = read_lane_f64(im0, 0), a10 = read_lane_f64(im0, 1), a11 = read_lane_f64(im1, 1);
            const double l00 = sqrt(a00);
            const double l10 = a10 / l00;
            const double l11 = sqrt(a11 - l10 * l10);
            const double s00 = 1.0 / l00;
            const double s01 = (0.0 - l10 * s00) / l11;
            const double s11 = 1.0 / l11;
            sm0 = lane == 0 ? s00 : 0.0;
            sm1 = lane == 0 ? s01 : s11;
            wc_n = 0;
            adapt_da_restart(da, step_size);
          }
        } else {
        if (stage != 0) {
          wc_n += 1;
          if (el) adapt_welford_elem(sq, wc_n, wmean, wm2);
        }
        if (wend) {
          if (el) adapt_window_end_elem(wc_n, wmean, wm2, imr, smr);
          wc_n = 0;
          adapt_da_restart(da, step_size);
        }
        }
        if (t_idx == m.T - 1) step_size = exp(da.x_avg);  // window_adaptation.py:184-190
        eps = step_size;
      }
      t_idx += 1;
      if (t_idx == m.T) {
        fin = true;
        if (el) {
          a.q[c * 2 + lane] = sq;
          a.g[c * 2 + lane] = sg;
          if (a.out.momentum) a.out.momentum[c * 2 + lane] = sp;
          if (m.adapt && DM) {
            m.ad.wc_mean[c * 2 + lane] = wmean;
            m.ad.wc_m2[c * 4 + 2 * lane] = wm2_0;
            m.ad.wc_m2[c * 4 + 2 * lane + 1] = wm2_1;
            m.ad.imm[c * 4 + 2 * lane] = im0;
            m.ad.imm[c * 4 + 2 * lane + 1] = im1;
            m.ad.sqrt_mass[c * 4 + 2 * lane] = sm0;
            m.ad.sqrt_mass[c * 4 + 2 * lane + 1] = sm1;
          } else if (m.adapt) {
            m.ad.wc_mean[c * 2 + lane] = wmean;
            m.ad.wc_m2[c * 2 + lane] = wm2;
            m.ad.imm[c * 2 + lane] = imr;
            m.ad.sqrt_mass[c * 2 + lane] = smr;
          }
        }
        if (lane == 0) {
          a.U[c] = U_state;
          a.out.acceptance_probability[c] = ct.acc_prob;
          if (a.out.num_doublings) a.out.num_doublings[c] = ct.ndoubl;
          if (a.out.is_turning) a.out.is_turning[c] = ct.out_turn;
          a.out.is_diverging[c] = ct.out_div;
          if (a.out.n_leapfrog) a.out.n_leapfrog[c] = ct.nleap;
          if (m.nleap_total) m.nleap_total[c] = nleap_sum;
          if (m.adapt) {
            m.ad.da_step[c] = da.step;
            m.ad.da_x[c] = da.x;
            m.ad.da_x_avg[c] = da.x_avg;
            m.ad.da_g_avg[c] = da.g_avg;
            m.ad.da_mu[c] = da.mu;
            m.ad.wc_n[c] = wc_n;
            m.ad.step_size[c] = eps;
          }
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (k < a.nsites) pcg_store(a.rng + ((size_t)c * a.nsites + k) * 4, rng.g[k]);
        }
      } else {
        begin_transition();
      }
    }
    LRN_TICK(5);  // transition end / begin
  }
#ifdef AEHMC_WIDE_TIMING
  if (lane == 0 && c < a.C)
    for (int k = 0; k < 8; k++) a.ckp[c * 8 + k] = (double)tacc[k];
  if (lane == 0 && wave >= 4 && (long long)blockIdx.x * 4 + wave - 4 < a.C)  // the row-serving waves
    for (int k = 0; k < 8; k++) a.cks[((long long)blockIdx.x * 4 + wave - 4) * 8 + k] = (double)tacc[k];
#endif
}

inline bool nuts_linreg_supported(int tkind, int met_ndim, long long D, long long max_exp) {
  return tkind == AEHMC_T_LINREG && met_ndim <= 2 && D == 2 && max_exp <= NUTS_LINREG_MAX_EXP;
}

inline hipError_t launch_nuts_linreg(const EngineArgs &a, const NutsSampleArgs &m, hipStream_t st) {
  const unsigned grid = (unsigned)((a.C + 3) / 4);
  const size_t dyn = (size_t)2 * (a.N <= NUTS_LINREG_LDS_ROWS ? a.N : NUTS_LINREG_LDS_ROWS) * sizeof(double);
  if (a.met_ndim == 2) {
    if (m.adapt && !(m.ad.full && a.imm_cs)) return hipErrorInvalidValue;  // (adaptation: one matrix per chain)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_linreg<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_nuts_linreg<true>, dim3(grid), dim3(LR_BLOCK), dyn, st, a, m);
    return hipGetLastError();
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nuts_linreg<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_nuts_linreg<false>, dim3(grid), dim3(LR_BLOCK), dyn, st, a, m);
  return hipGetLastError();
}

}  // namespace aehmc
