// The scalar state machine of one NUTS transition (round 5): the per-chain decisions of dynamic_integration and
// multiplicative_expansion -- reference aehmc/trajectory.py:195-336 (one scan step), 516-608 (expand_once),
// proposals.py:19-174, termination.py:85-131 -- written ONCE and used by every kernel family: the lock-step engine and
// the kernels built from its device functions (engine.cuh: nuts_book / nuts_finalize_expansion; k_nuts_pc_dense,
// k_nuts_block_dense), the register-resident kernels (nuts_resident.cuh), the workgroup-per-chain kernel
// (nuts_wide.cuh), the regression kernel (nuts_linreg.cuh) and the block-resident kernels (nuts_block_tree.inc).
// What differs between the families -- where a chain's vectors live (global rows, LDS rows, registers), how its sums
// are reduced, where its generators are parked -- stays with them; they hand this header the reduced scalars and a
// functor that draws binomial(1, p) from the right call site.
//
// LANES: one wavefront (or more) owns the chain, so the independent transcendental chains of a step are evaluated in
// different LANES at once (same instruction sequence per lane as the scalar expressions: same bits).  Sub-wavefront
// teams (several chains per wavefront: k_nuts_resident below 64 lanes) use the scalar form.
//
// Included by engine.cuh behind ChainCtl, np_logaddexp, read_lane_f64 and the generators.
#pragma once

namespace aehmc {

// The per-leapfrog scalars of dynamic_integration (proposals.py:96-99, 141-144): the acceptance
// probability expit(w_new - w_sub) and the two running logaddexp's.  All inputs are wave-uniform
// and the three results are independent given the new weight, so they are evaluated in three
// LANES of the wave at once -- one vector exp and one vector log1p instead of three and two
// scalar ones (the 64-wide redundant evaluation of the same scalar costs exactly as much as a
// lane-varying one).  Each lane runs the very instruction sequence of np_logaddexp / the scalar
// expression, so the results are the same bits.  Valid when one wavefront (or more) owns the chain.
struct StepScalars {
  double pa, sub_w, sub_slpa;
};
__device__ __forceinline__ StepScalars nuts_step_scalars(double sub_w, double np_w, double sub_slpa,
                                                         double np_slpa, int lane) {
  const double x = lane == 1 ? sub_w : sub_slpa, y = lane == 1 ? np_w : np_slpa;  // lanes 1, 2: logaddexp(x, y)
  const double tmp = x - y;
  const double earg = lane == 0 ? -(np_w - sub_w) : (tmp > 0 ? -tmp : tmp);
  const double e = exp(earg);
  const double l = log1p(e);
  double la = (x == y) ? x + 0.693147180559945309417232121458176568 : (tmp > 0 ? x + l : (tmp <= 0 ? y + l : tmp));
  double pa = 1.0 / (1.0 + e);
  if (isnan(pa)) pa = 0.0;
  const double r = lane == 0 ? pa : la;
  StepScalars o;
  o.pa = read_lane_f64(r, 0);
  o.sub_w = read_lane_f64(r, 1);
  o.sub_slpa = read_lane_f64(r, 2);
  return o;
}

// The scalars at the end of a sub-trajectory (trajectory.py:537-608, proposals.py:105-174): the
// acceptance statistic exp(slpa_sub), the biased-sampling ratio exp(w_sub - w_prop) and the two
// logaddexp's that merge the sub-trajectory into the proposal, in four lanes at once.  `swap`
// (a diverged or U-turned sub-trajectory) only changes the argument order of the slpa merge.
struct ExpansionScalars {
  double e_slpa, e_ratio, la_w, la_slpa;
};
__device__ __forceinline__ ExpansionScalars nuts_expansion_scalars(double sub_w, double prop_w, double sub_slpa,
                                                                   double prop_slpa, bool swap, int lane) {
  const double x = lane == 2 ? prop_w : (swap ? sub_slpa : prop_slpa);   // lanes 2, 3: logaddexp(x, y)
  const double y = lane == 2 ? sub_w : (swap ? prop_slpa : sub_slpa);
  const double tmp = x - y;
  const double earg = lane == 0 ? sub_slpa : lane == 1 ? sub_w - prop_w : (tmp > 0 ? -tmp : tmp);
  const double e = exp(earg);
  const double l = log1p(e);
  const double la = (x == y) ? x + 0.693147180559945309417232121458176568 : (tmp > 0 ? x + l : (tmp <= 0 ? y + l : tmp));
  const double r = lane < 2 ? e : la;
  ExpansionScalars o;
  o.e_slpa = read_lane_f64(r, 0);
  o.e_ratio = read_lane_f64(r, 1);
  o.la_w = read_lane_f64(r, 2);
  o.la_slpa = read_lane_f64(r, 3);
  return o;
}

// ---- termination.py:85-131: which checkpoints step `step` of a sub-trajectory compares against.  Step 0 inherits the
// (stale) indices of the previous sub-trajectory (termination.py:109-113).
struct TreeIdx {
  int tmin, tmax;
};
__device__ __forceinline__ TreeIdx tree_step_indices(int step, int stale_min, int stale_max) {
  TreeIdx o;
  if (step == 0) {
    o.tmin = stale_min;
    o.tmax = stale_max;
  } else {
    const int n1 = __ffs(~step) - 1;
    o.tmax = __popc(step >> 1);
    o.tmin = o.tmax - n1 + 1;
  }
  return o;
}

// ---- proposals.py:19-62: energy, weight, log acceptance statistic and divergence of the new point
struct TreePoint {
  double E, w, slpa;
  bool div;
};
__device__ __forceinline__ TreePoint tree_new_point(double H0, double U, double kd, double thr) {
  TreePoint o;
  o.E = U + 0.5 * kd;
  double delta = H0 - o.E;
  if (isnan(delta)) delta = -INFINITY;
  o.div = fabs(delta) > thr;
  o.w = delta;
  o.slpa = delta > 0 ? 0.0 : delta;
  return o;
}

// ---- trajectory.py:262-305 + proposals.py:72-102, 141-144: the sub-trajectory's proposal after one more point.
// Returns whether the sub-trajectory proposal becomes the new point (the caller copies the vectors).
// `draw(p)` = binomial(1, p) from call site #3 (progressive_uniform_sampling).
template <bool LANES, class Draw>
__device__ __forceinline__ bool tree_sample_step(ChainCtl &ct, int step, const TreePoint &np, int lane, Draw draw) {
  if (step == 0) {
    ct.sub_E = np.E;
    ct.sub_w = np.w;
    ct.sub_slpa = np.slpa;
    ct.length = 1;
    return true;
  }
  int acc;
  if (LANES) {
    const StepScalars sc = nuts_step_scalars(ct.sub_w, np.w, ct.sub_slpa, np.slpa, lane);
    acc = draw(sc.pa);
    ct.sub_w = sc.sub_w;
    ct.sub_slpa = sc.sub_slpa;
  } else {
    double pa = 1.0 / (1.0 + exp(-(np.w - ct.sub_w)));  // proposals.py:96-99
    if (isnan(pa)) pa = 0.0;
    acc = draw(pa);
    ct.sub_w = np_logaddexp(ct.sub_w, np.w);
    ct.sub_slpa = np_logaddexp(ct.sub_slpa, np.slpa);
  }
  ct.length += 1;
  if (acc) {
    ct.sub_E = np.E;
    return !ct.phantom;
  }
  return false;
}

// ---- trajectory.py:289-336, 537-545: what the scan does behind this step.  A first step that diverged ends the
// expansion at once, yet the reference's scan still runs its 2**j remaining steps and draws from site #3 (SURVEY quirk:
// the caller turns the chain into a phantom); a phantom that reaches the end of its scan is done.
struct TreeControl {
  bool finalize, fin_div, fin_term;
};
__device__ __forceinline__ TreeControl tree_step_control(ChainCtl &ct, int step, bool div, bool term) {
  TreeControl o = {false, false, false};
  if (step == 0 && div && !ct.phantom) {
    o.finalize = true;
    o.fin_div = true;
  } else if (step >= 1 && (div || term || step == (1 << ct.j))) {
    if (ct.phantom) ct.done = 1;
    else {
      o.finalize = true;
      o.fin_div = div;
      o.fin_term = term;
    }
  } else {
    ct.step = step + 1;
  }
  return o;
}

// ---- trajectory.py:551-564, proposals.py:105-174: the finished sub-trajectory merged into the transition's proposal.
// Returns whether the main proposal becomes the sub-trajectory's (progressive_biased_sampling); the caller swaps its
// proposal slots.  `draw(p)` = binomial(1, p) from call site #4 -- always drawn (proposals.py:130).
template <bool LANES, class Draw>
__device__ __forceinline__ bool tree_merge_expansion(ChainCtl &ct, bool fin_div, bool fin_term, int lane, Draw draw) {
  const bool keep = fin_div || fin_term;  // a diverged / U-turned sub-trajectory never replaces the proposal
  double e_slpa, e_ratio, la_w = ct.prop_w, la_slpa;
  if (LANES) {
    const ExpansionScalars es = nuts_expansion_scalars(ct.sub_w, ct.prop_w, ct.sub_slpa, ct.prop_slpa, keep, lane);
    e_slpa = es.e_slpa;
    e_ratio = es.e_ratio;
    la_w = es.la_w;
    la_slpa = es.la_slpa;
  } else {
    e_slpa = exp(ct.sub_slpa);
    e_ratio = exp(ct.sub_w - ct.prop_w);
    if (keep) {
      la_slpa = np_logaddexp(ct.sub_slpa, ct.prop_slpa);  // trajectory.py:560-564
    } else {
      la_w = np_logaddexp(ct.prop_w, ct.sub_w);           // proposals.py:141-144
      la_slpa = np_logaddexp(ct.prop_slpa, ct.sub_slpa);
    }
  }
  ct.acc_prob = e_slpa / (double)ct.length;  // trajectory.py:551-553
  double pbias = e_ratio;                    // proposals.py:130
  if (pbias > 1.0) pbias = 1.0;
  if (pbias < 0.0) pbias = 0.0;
  const int acc_b = draw(pbias);
  ct.prop_slpa = la_slpa;
  if (keep) return false;
  ct.prop_w = la_w;
  if (acc_b) ct.prop_E = ct.sub_E;
  return acc_b != 0;
}

// ---- trajectory.py:566-608: the expansion's record and whether the transition ends with it
__device__ __forceinline__ bool tree_expansion_outcome(ChainCtl &ct, bool fin_div, bool fin_term, bool turning, int max_exp) {
  ct.ndoubl = ct.j + 1;
  ct.out_div = fin_div;
  ct.out_turn = turning;
  return fin_div || turning || fin_term || (ct.j + 1 == max_exp);
}

}  // namespace aehmc
