/* CPU restatement (plain C) of aehmc's HMC / NUTS transition.  TEST INFRASTRUCTURE ONLY.
 *
 * Parity oracle + timed CPU baseline ("port").  Loaded (ctypes) only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
 * package.  Pinned against the reference's published values G1/G2 and its unit-test
 * tables in tests/test_oracle_golden.py, and against oracle/np_oracle.py.
 *
 * Each function cites the reference file:line (under /root/reference) it follows.
 * PARITY UNPINNED (no published reference value): dense-metric trajectories, and RNG
 * consumption after a sub-trajectory whose first step diverged (trajectory.py:336).
 * The dense branch is tied to the pinned diagonal branch by an exact invariance instead:
 * a transition is equivariant under q' = A q for lower-triangular A
 * (tests/test_oracle_golden.py::test_dense_branch_equals_diagonal_branch_under_triangular_map).
 * The RNG restates numpy 2.2.6's PCG64 / random_standard_normal / random_binomial
 * (third party; reached by the reference through aesara RandomStream, "scheme A":
 * one spawned SeedSequence child per RNG call site) and is checked bit-for-bit
 * against numpy.random.Generator in tests/test_rng_parity.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aehmc_ziggurat_tables.h"

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ RNG ---------- */
typedef struct { uint64_t state_hi, state_lo, inc_hi, inc_lo; } ao_pcg64;

static const uint64_t ki_double[256] = { AEHMC_ZIG_KI_VALUES };
static const double wi_double[256] = { AEHMC_ZIG_WI_VALUES };
static const double fi_double[256] = { AEHMC_ZIG_FI_VALUES };

#define PCG_MULT ((((u128)2549297995355413924ULL) << 64) | (u128)4865540595714422341ULL)

static inline uint64_t pcg64_next64(ao_pcg64 *r) {
  u128 s = (((u128)r->state_hi) << 64) | r->state_lo;
  u128 inc = (((u128)r->inc_hi) << 64) | r->inc_lo;
  s = s * PCG_MULT + inc;                       /* 128-bit LCG step */
  r->state_hi = (uint64_t)(s >> 64);
  r->state_lo = (uint64_t)s;
  uint64_t x = r->state_hi ^ r->state_lo;       /* XSL-RR output on the new state */
  unsigned rot = (unsigned)(r->state_hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
static inline double pcg64_next_double(ao_pcg64 *r) {
  return (double)(pcg64_next64(r) >> 11) * (1.0 / 9007199254740992.0);
}

/* numpy random_standard_normal (ziggurat, 256 layers) */
static double rng_standard_normal(ao_pcg64 *rng) {
  for (;;) {
    uint64_t r = pcg64_next64(rng);
    int idx = (int)(r & 0xff);
    r >>= 8;
    int sign = (int)(r & 0x1);
    uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * wi_double[idx];
    if (sign & 0x1) x = -x;
    if (rabs < ki_double[idx]) return x;
    if (idx == 0) {
      for (;;) {
        double xx = -AEHMC_ZIG_NOR_INV_R * log1p(-pcg64_next_double(rng));
        double yy = -log1p(-pcg64_next_double(rng));
        if (yy + yy > xx * xx)
          return ((rabs >> 8) & 0x1) ? -(AEHMC_ZIG_NOR_R + xx) : AEHMC_ZIG_NOR_R + xx;
      }
    } else {
      if (((fi_double[idx - 1] - fi_double[idx]) * pcg64_next_double(rng) + fi_double[idx]) <
          exp(-0.5 * x * x))
        return x;
    }
  }
}

/* numpy random_binomial_inversion specialised to n == 1 (literal loop kept) */
static int64_t binomial1_inversion(ao_pcg64 *rng, double p) {
  double q = 1.0 - p;
  double qn = exp(1 * log(q));
  double np_ = 1 * p;
  double b = np_ + 10.0 * sqrt(np_ * q + 1);
  int64_t bound = (int64_t)(1.0 < b ? 1.0 : b);
  int64_t X = 0;
  double px = qn;
  double U = pcg64_next_double(rng);
  while (U > px) {
    X++;
    if (X > bound) {
      X = 0;
      px = qn;
      U = pcg64_next_double(rng);
    } else {
      U -= px;
      px = ((1 - X + 1) * p * px) / (X * q);
    }
  }
  return X;
}
/* Generator.binomial(1, p): aesara bernoulli -> scipy bernoulli.rvs -> this */
static int rng_bernoulli(ao_pcg64 *rng, double p) {
  if (p == 0.0) return 0;                       /* no draw */
  if (p <= 0.5) return (int)binomial1_inversion(rng, p);
  return (int)(1 - binomial1_inversion(rng, 1.0 - p));
}

void ao_rng_normals(ao_pcg64 *rng, int64_t n, double *out) {
  for (int64_t i = 0; i < n; i++) out[i] = rng_standard_normal(rng);
}
void ao_rng_bernoulli(ao_pcg64 *rng, int64_t n, const double *p, int32_t *out) {
  for (int64_t i = 0; i < n; i++) out[i] = rng_bernoulli(rng, p[i]);
}
void ao_rng_doubles(ao_pcg64 *rng, int64_t n, double *out) {
  for (int64_t i = 0; i < n; i++) out[i] = pcg64_next_double(rng);
}

/* ------------------------------------------------------------------ targets ------ */
enum { AO_T_STD_NORMAL = 0, AO_T_ISO_GAUSSIAN = 1, AO_T_DIAG_GAUSSIAN = 2,
       AO_T_DENSE_MVN = 3, AO_T_LINREG = 4 };

typedef struct {
  int32_t kind;
  int32_t pad;
  int64_t D;
  const double *mu;     /* diag / dense */
  const double *sigma;  /* diag */
  const double *prec;   /* dense, row-major [D,D] */
  const double *X;      /* linreg */
  const double *y;
  int64_t N;
} ao_target;

#define LOG_SQRT_2PI 0.91893853320467267 /* == np.log(np.sqrt(2*np.pi)) */

/* potential U = -logprob and its gradient (hmc.py:16-40, integrators.py:64-65) */
static double target_eval(const ao_target *t, const double *q, double *g, double *scratch) {
  int64_t D = t->D;
  double u = 0.0;
  switch (t->kind) {
  case AO_T_STD_NORMAL:
    for (int64_t i = 0; i < D; i++) { u += 0.5 * (q[i] * q[i]) + LOG_SQRT_2PI; g[i] = q[i]; }
    return u;
  case AO_T_ISO_GAUSSIAN:
    for (int64_t i = 0; i < D; i++) { u += q[i] * q[i]; g[i] = q[i]; }
    return 0.5 * u;
  case AO_T_DIAG_GAUSSIAN:
    for (int64_t i = 0; i < D; i++) {
      double z = (q[i] - t->mu[i]) / t->sigma[i];
      u += 0.5 * (z * z) + log(t->sigma[i]) + LOG_SQRT_2PI;
      g[i] = z / t->sigma[i];
    }
    return u;
  case AO_T_DENSE_MVN: {
    double *r = scratch;
    for (int64_t i = 0; i < D; i++) r[i] = q[i] - t->mu[i];
    for (int64_t i = 0; i < D; i++) {
      const double *row = t->prec + i * D;
      double acc = 0.0;
      for (int64_t k = 0; k < D; k++) acc += row[k] * r[k];
      g[i] = acc;
    }
    for (int64_t i = 0; i < D; i++) u += r[i] * g[i];
    return 0.5 * u;
  }
  case AO_T_LINREG: { /* examples/LinearRegression.ipynb:126-166; q = [w, log n] */
    double w = q[0], ell = q[1], n = exp(ell), n2 = n * n;
    double s_xr = 0.0, s_rr = 0.0;
    for (int64_t i = 0; i < t->N; i++) {
      double r = t->y[i] - t->X[i] * w;
      s_xr += t->X[i] * r;
      s_rr += r * r;
    }
    double N = (double)t->N;
    double lp_w = -0.5 * w * w - LOG_SQRT_2PI;
    double lp_n = log(n) - n + ell;
    double lp_y = -0.5 * (s_rr / n2) - N * LOG_SQRT_2PI - N * ell;
    g[0] = -(-w + s_xr / n2);
    g[1] = -(2.0 - n - N + s_rr / n2);
    return -(lp_w + lp_n + lp_y);
  }
  }
  return NAN;
}

/* ------------------------------------------------------------------ metric ------- */
typedef struct {
  int32_t ndim;            /* 0 scalar, 1 diagonal, 2 dense (metrics.py:44-63) */
  int32_t pad;
  int64_t D;
  const double *imm;       /* [1] | [D] | [D,D] row-major */
  const double *sqrt_mass; /* sqrt(1/imm) [1]|[D], or L^-T [D,D] (metrics.py:45,49,58) */
} ao_metric;

static void metric_velocity(const ao_metric *m, const double *p, double *v) {
  int64_t D = m->D;
  if (m->ndim == 0) { for (int64_t i = 0; i < D; i++) v[i] = m->imm[0] * p[i]; }
  else if (m->ndim == 1) { for (int64_t i = 0; i < D; i++) v[i] = m->imm[i] * p[i]; }
  else {
    for (int64_t i = 0; i < D; i++) {
      const double *row = m->imm + i * D;
      double acc = 0.0;
      for (int64_t k = 0; k < D; k++) acc += row[k] * p[k];
      v[i] = acc;
    }
  }
}
static double dotd(const double *a, const double *b, int64_t D) {
  double s = 0.0;
  for (int64_t i = 0; i < D; i++) s += a[i] * b[i];
  return s;
}
/* metrics.py:70-73 */
static double metric_kinetic(const ao_metric *m, const double *p, double *v) {
  metric_velocity(m, p, v);
  return 0.5 * dotd(v, p, m->D);
}
/* metrics.py:65-68 */
static void metric_momentum(const ao_metric *m, ao_pcg64 *rng, double *p, double *z) {
  int64_t D = m->D;
  for (int64_t i = 0; i < D; i++) z[i] = rng_standard_normal(rng);
  if (m->ndim == 0) { for (int64_t i = 0; i < D; i++) p[i] = m->sqrt_mass[0] * z[i]; }
  else if (m->ndim == 1) { for (int64_t i = 0; i < D; i++) p[i] = m->sqrt_mass[i] * z[i]; }
  else {
    for (int64_t i = 0; i < D; i++) {
      const double *row = m->sqrt_mass + i * D;
      double acc = 0.0;
      for (int64_t k = 0; k < D; k++) acc += row[k] * z[k];
      p[i] = acc;
    }
  }
}
/* metrics.py:75-104; scratch: 3*D */
static int metric_is_turning(const ao_metric *m, const double *pl, const double *pr,
                             const double *psum, double *scratch) {
  int64_t D = m->D;
  double *vl = scratch, *vr = scratch + D, *rho = scratch + 2 * D;
  metric_velocity(m, pl, vl);
  metric_velocity(m, pr, vr);
  for (int64_t i = 0; i < D; i++) rho[i] = psum[i] - (pr[i] + pl[i]) / 2;
  return (dotd(vl, rho, D) <= 0) | (dotd(vr, rho, D) <= 0);
}

/* ------------------------------------------------------------------ integrator --- */
typedef struct { double *q, *p, *g; double U; } ao_state;

/* integrators.py:54-73 (in place) */
static void leapfrog(const ao_target *t, const ao_metric *m, ao_state *s, double step_size,
                     double *v, double *scratch) {
  int64_t D = t->D;
  double b1e = 0.5 * step_size, a2e = 1 * step_size;
  for (int64_t i = 0; i < D; i++) s->p[i] = s->p[i] - b1e * s->g[i];
  metric_velocity(m, s->p, v);
  for (int64_t i = 0; i < D; i++) s->q[i] = s->q[i] + a2e * v[i];
  s->U = target_eval(t, s->q, s->g, scratch);
  for (int64_t i = 0; i < D; i++) s->p[i] = s->p[i] - b1e * s->g[i];
}

static void state_copy(ao_state *dst, const ao_state *src, int64_t D) {
  memcpy(dst->q, src->q, D * sizeof(double));
  memcpy(dst->p, src->p, D * sizeof(double));
  memcpy(dst->g, src->g, D * sizeof(double));
  dst->U = src->U;
}
static void state_alloc(ao_state *s, int64_t D) {
  s->q = (double *)malloc(3 * D * sizeof(double));
  s->p = s->q + D;
  s->g = s->q + 2 * D;
  s->U = 0;
}

static double np_logaddexp(double x, double y) { /* numpy npy_logaddexp */
  if (x == y) return x + 0.693147180559945309417232121458176568;
  double tmp = x - y;
  if (tmp > 0) return x + log1p(exp(-tmp));
  if (tmp <= 0) return y + log1p(exp(tmp));
  return tmp;
}

/* new_state (hmc.py:16-40) */
void ao_new_state(const ao_target *t, int64_t C, const double *q, double *U, double *g) {
  int64_t D = t->D;
#pragma omp parallel
  {
    double *scratch = (double *)malloc(D * sizeof(double));
#pragma omp for
    for (int64_t c = 0; c < C; c++) U[c] = target_eval(t, q + c * D, g + c * D, scratch);
    free(scratch);
  }
}

/* ------------------------------------------------------------------ HMC ---------- */
/* hmc.py:77-124 + hmc.py:157-204 + trajectory.py:79-105; rng[0]=momentum, rng[1]=accept */
static void hmc_step_one(const ao_target *t, const ao_metric *m, ao_pcg64 *rng, double eps,
                         int64_t L, double thr, double *q, double *U, double *g, double *p_out,
                         double *acc_prob, int32_t *is_div, int32_t *accepted) {
  int64_t D = t->D;
  ao_state s;
  state_alloc(&s, D);
  double *v = (double *)malloc(5 * D * sizeof(double));
  double *scratch = v + D, *p0 = v + 2 * D, *z = v + 3 * D;
  metric_momentum(m, &rng[0], p0, z);
  memcpy(s.q, q, D * sizeof(double));
  memcpy(s.p, p0, D * sizeof(double));
  memcpy(s.g, g, D * sizeof(double));
  s.U = *U;
  for (int64_t l = 0; l < L; l++) leapfrog(t, m, &s, eps, v, scratch);
  for (int64_t i = 0; i < D; i++) s.p[i] = -1.0 * s.p[i];
  double energy = *U + metric_kinetic(m, p0, v);
  double new_energy = s.U + metric_kinetic(m, s.p, v);
  double delta = energy - new_energy;
  if (isnan(delta)) delta = -INFINITY;
  *is_div = fabs(delta) > thr;
  double pa = exp(delta);
  if (pa > 1.0) pa = 1.0;
  if (pa < 0.0) pa = 0.0;
  *acc_prob = pa;
  int acc = rng_bernoulli(&rng[1], pa);
  *accepted = acc;
  if (acc) {
    memcpy(q, s.q, D * sizeof(double));
    memcpy(g, s.g, D * sizeof(double));
    memcpy(p_out, s.p, D * sizeof(double));
    *U = s.U;
  } else {
    memcpy(p_out, p0, D * sizeof(double));
  }
  free(v);
  free(s.q);
}

void ao_hmc_step(const ao_target *t, const ao_metric *m, int64_t C, ao_pcg64 *rng /*[C][2]*/,
                 double eps, int64_t L, double thr, double *q, double *U, double *g,
                 double *p_out, double *acc_prob, int32_t *is_div, int32_t *accepted,
                 int32_t nthreads) {
  int64_t D = t->D;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for (int64_t c = 0; c < C; c++)
    hmc_step_one(t, m, rng + 2 * c, eps, L, thr, q + c * D, U + c, g + c * D, p_out + c * D,
                 acc_prob + c, is_div + c, accepted + c);
}

/* ------------------------------------------------------------------ NUTS --------- */
typedef struct { double E, w, slpa; } ao_prop; /* proposals.py:11-15 scalars */

/* termination.py:192-235 literal loops */
static void find_storage_indices(int64_t step, int64_t *idx_min, int64_t *idx_max) {
  int64_t nc0 = step, nc1 = -1;
  for (int64_t it = 0; it < step + 1; it++) {
    int stop = (nc0 & 1) == 0;
    nc0 = nc0 / 2; nc1 = nc1 + 1;
    if (stop) break;
  }
  int64_t num_subtrees = nc1;
  nc0 = step / 2; nc1 = 0;
  for (int64_t it = 0; it < step + 1; it++) {
    int stop = nc0 == 0;
    nc1 = nc1 + (nc0 & 1); nc0 = nc0 / 2;
    if (stop) break;
  }
  *idx_max = nc1;
  *idx_min = nc1 - num_subtrees + 1;
}
void ao_find_storage_indices(int64_t step, int64_t *mn, int64_t *mx) { find_storage_indices(step, mn, mx); }

typedef struct { double *ckp, *cks; int64_t mn, mx; } ao_term; /* termination.py:12-16 */

/* EXPERIMENT switches (tests/test_nuts_quirks.py only; both 0 = the reference's semantics,
 * which is what every parity test pins).  They exist to attribute the stationary bias of the
 * reference's NUTS to one of its two literal quirks:
 *   alt_quirk1 = 1: a sub-trajectory of expansion j takes 2**j leapfrogs instead of the
 *                   reference's 2**j + 1 (trajectory.py:307 scans n_steps = max_num_steps AFTER the
 *                   first step of trajectory.py:276-284);
 *   alt_quirk2 = 1: step 0 of a sub-trajectory uses _find_storage_indices(0) instead of the
 *                   indices inherited from the previous sub-trajectory (termination.py:109-113). */
static int alt_quirk1 = 0, alt_quirk2 = 0;
void ao_set_experiment(int32_t q1, int32_t q2) { alt_quirk1 = q1; alt_quirk2 = q2; }

/* termination.py:85-131 */
static void term_update(ao_term *T, const double *psum, const double *p, int64_t step, int64_t D) {
  int64_t mn, mx;
  if (step == 0 && !alt_quirk2) { mn = T->mn; mx = T->mx; }     /* inherited, possibly stale */
  else find_storage_indices(step, &mn, &mx);
  if (step % 2 == 0) {
    memcpy(T->ckp + mx * D, p, D * sizeof(double));
    memcpy(T->cks + mx * D, psum, D * sizeof(double));
  }
  T->mn = mn; T->mx = mx;
}
/* termination.py:133-187; scratch 4*D */
static int term_is_turning(const ao_metric *m, const ao_term *T, const double *psum,
                           const double *p, double *scratch) {
  int64_t D = m->D;
  if (T->mx < T->mn) return 0;
  double *sub = scratch;
  int crit = 0;
  int64_t i = T->mx;
  for (int64_t it = 0; it < T->mx + 2; it++) {
    const double *ck = T->ckp + i * D, *cs = T->cks + i * D;
    for (int64_t k = 0; k < D; k++) sub[k] = psum[k] - cs[k] + ck[k];
    crit = metric_is_turning(m, ck, p, sub, scratch + D);
    int reached = (i - 1) < T->mn;
    i = i - 1;
    if (crit || reached) break;
  }
  return crit;
}
int ao_is_iterative_turning(const ao_metric *m, int64_t max_exp, const double *ckp,
                            const double *cks, int64_t mn, int64_t mx, const double *psum,
                            const double *p) {
  ao_term T = { (double *)ckp, (double *)cks, mn, mx };
  double *scratch = (double *)malloc(4 * m->D * sizeof(double));
  int r = term_is_turning(m, &T, psum, p, scratch);
  free(scratch);
  (void)max_exp;
  return r;
}
int ao_is_turning(const ao_metric *m, const double *pl, const double *pr, const double *ps) {
  double *scratch = (double *)malloc(3 * m->D * sizeof(double));
  int r = metric_is_turning(m, pl, pr, ps, scratch);
  free(scratch);
  return r;
}
double ao_kinetic_energy(const ao_metric *m, const double *p) {
  double *v = (double *)malloc(m->D * sizeof(double));
  double k = metric_kinetic(m, p, v);
  free(v);
  return k;
}

/* proposals.py:19-62 */
static int gen_proposal(const ao_metric *m, double H0, const ao_state *s, double thr,
                        ao_prop *out, double *v) {
  double E = s->U + metric_kinetic(m, s->p, v);
  double delta = H0 - E;
  if (isnan(delta)) delta = -INFINITY;
  out->E = E;
  out->w = delta;
  out->slpa = delta > 0 ? 0.0 : delta;
  return fabs(delta) > thr;
}

/* nuts.py:56-153 + trajectory.py:428-714 + trajectory.py:154-374.
 * rng[0]=momentum (#1), rng[1]=direction (#2), rng[2]=uniform progressive (#3),
 * rng[3]=biased progressive (#4). */
static void nuts_step_one(const ao_target *t, const ao_metric *m, ao_pcg64 *rng, double eps,
                          int64_t max_exp, double thr, double *q, double *U, double *g,
                          double *p_out, double *acc_prob, int64_t *num_doublings,
                          int32_t *is_turning_out, int32_t *is_div_out, int64_t *n_leapfrog) {
  int64_t D = t->D;
  ao_state left, right, prop_st, sub_st, init_st;
  state_alloc(&left, D); state_alloc(&right, D); state_alloc(&prop_st, D);
  state_alloc(&sub_st, D); state_alloc(&init_st, D);
  double *buf = (double *)calloc((size_t)(10 * D + 2 * max_exp * D), sizeof(double));
  double *v = buf, *scratch = buf + D /*5D*/, *psum = buf + 6 * D, *psum_sub = buf + 7 * D,
         *z = buf + 8 * D, *init_psum = buf + 9 * D;
  ao_term T = { buf + 10 * D, buf + 10 * D + max_exp * D, 0, 0 }; /* termination.py:43-83 */

  /* nuts.py:113-125 */
  metric_momentum(m, &rng[0], left.p, z);
  memcpy(left.q, q, D * sizeof(double));
  memcpy(left.g, g, D * sizeof(double));
  left.U = *U;
  state_copy(&right, &left, D);
  state_copy(&prop_st, &left, D);
  double H0 = left.U + metric_kinetic(m, left.p, v);
  ao_prop prop = { H0, 0.0, -INFINITY };
  memcpy(psum, left.p, D * sizeof(double));
  int64_t nleap = 0;

  double acc = 0.0; int64_t nd = 0; int out_div = 0, out_turn = 0;
  for (int64_t j = 0; j < max_exp; j++) {               /* trajectory.py:463-608 */
    int go_right = rng_bernoulli(&rng[1], 0.5);          /* trajectory.py:516 */
    double d = go_right ? 1.0 : -1.0;
    ao_state *s = go_right ? &right : &left;             /* integrate in place on that end */
    double step_size = d * eps;
    int64_t max_num_steps = ((int64_t)1 << j) - (alt_quirk1 ? 1 : 0);

    /* ---- dynamic_integration.integrate: first step, trajectory.py:276-305 ---- */
    leapfrog(t, m, s, step_size, v, scratch); nleap++;
    ao_prop sub;                                          /* subtree proposal scalars */
    int div0 = gen_proposal(m, H0, s, thr, &sub, v);
    state_copy(&sub_st, s, D);
    memcpy(psum_sub, s->p, D * sizeof(double));
    term_update(&T, psum_sub, s->p, 0, D);
    int64_t length = 1; int is_div = div0, has_term = 0;
    /* the initial tuple is what integrate() returns when the first step diverged
     * (trajectory.py:336); the scan below still executes (and draws RNG) */
    ao_prop init_sub = sub; int64_t init_mn = T.mn, init_mx = T.mx;
    if (div0) { state_copy(&init_st, s, D); memcpy(init_psum, psum_sub, D * sizeof(double)); }

    /* ---- scan over steps 1..2**j with until: trajectory.py:307-332, 195-273 ---- */
    for (int64_t step = 1; step <= max_num_steps; step++) {
      leapfrog(t, m, s, step_size, v, scratch); if (!div0) nleap++;
      ao_prop np_;
      is_div = gen_proposal(m, H0, s, thr, &np_, v);
      /* progressive_uniform_sampling proposals.py:72-102 */
      double pa = 1.0 / (1.0 + exp(-(np_.w - sub.w)));
      if (isnan(pa)) pa = 0.0;
      int do_acc = rng_bernoulli(&rng[2], pa);
      sub.w = np_logaddexp(sub.w, np_.w);                 /* proposals.py:141-144 */
      sub.slpa = np_logaddexp(sub.slpa, np_.slpa);
      if (do_acc) { state_copy(&sub_st, s, D); sub.E = np_.E; }
      for (int64_t k = 0; k < D; k++) psum_sub[k] = psum_sub[k] + s->p[k];
      term_update(&T, psum_sub, s->p, step, D);
      has_term = term_is_turning(m, &T, psum_sub, s->p, scratch);
      length++;
      if (is_div || has_term) break;
    }
    if (div0) {                                           /* trajectory.py:336 */
      state_copy(s, &init_st, D); state_copy(&sub_st, &init_st, D);
      memcpy(psum_sub, init_psum, D * sizeof(double));
      sub = init_sub; T.mn = init_mn; T.mx = init_mx;     /* ckpt arrays unused afterwards */
      length = 1; is_div = 1; has_term = 0;
    }

    /* ---- back in expand_once: trajectory.py:537-608 ---- */
    for (int64_t k = 0; k < D; k++) psum[k] = psum[k] + psum_sub[k];
    acc = exp(sub.slpa) / (double)length;                 /* trajectory.py:551-553 */
    double pb = exp(sub.w - prop.w);                      /* proposals.py:130, drawn always */
    if (pb > 1.0) pb = 1.0;
    if (pb < 0.0) pb = 0.0;
    int acc_b = rng_bernoulli(&rng[3], pb);
    if (is_div || has_term) {
      prop.slpa = np_logaddexp(sub.slpa, prop.slpa);      /* trajectory.py:560-564 */
    } else {
      prop.w = np_logaddexp(prop.w, sub.w);
      prop.slpa = np_logaddexp(prop.slpa, sub.slpa);
      if (acc_b) { state_copy(&prop_st, &sub_st, D); prop.E = sub.E; }
    }
    int turning = metric_is_turning(m, left.p, right.p, psum, scratch);
    nd = j + 1; out_div = is_div; out_turn = turning;
    if (is_div || turning || has_term) break;
  }
  memcpy(q, prop_st.q, D * sizeof(double));
  memcpy(g, prop_st.g, D * sizeof(double));
  memcpy(p_out, prop_st.p, D * sizeof(double));
  *U = prop_st.U;
  *acc_prob = acc; *num_doublings = nd; *is_turning_out = out_turn; *is_div_out = out_div;
  *n_leapfrog = nleap;
  free(buf);
  free(left.q); free(right.q); free(prop_st.q); free(sub_st.q); free(init_st.q);
}

void ao_nuts_step(const ao_target *t, const ao_metric *m, int64_t C, ao_pcg64 *rng /*[C][4]*/,
                  double eps, int64_t max_exp, double thr, double *q, double *U, double *g,
                  double *p_out, double *acc_prob, int64_t *num_doublings, int32_t *is_turning,
                  int32_t *is_div, int64_t *n_leapfrog, int32_t nthreads) {
  int64_t D = t->D;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for (int64_t c = 0; c < C; c++)
    nuts_step_one(t, m, rng + 4 * c, eps, max_exp, thr, q + c * D, U + c, g + c * D,
                  p_out + c * D, acc_prob + c, num_doublings + c, is_turning + c, is_div + c,
                  n_leapfrog + c);
}

/* one leapfrog on C chains (for integrator known-answer tests) */
void ao_leapfrog(const ao_target *t, const ao_metric *m, int64_t C, double eps, int64_t nsteps,
                 double *q, double *p, double *U, double *g) {
  int64_t D = t->D;
  double *v = (double *)malloc(2 * D * sizeof(double));
  for (int64_t c = 0; c < C; c++) {
    ao_state s = { q + c * D, p + c * D, g + c * D, U[c] };
    for (int64_t l = 0; l < nsteps; l++) leapfrog(t, m, &s, eps, v, v + D);
    U[c] = s.U;
  }
  free(v);
}
