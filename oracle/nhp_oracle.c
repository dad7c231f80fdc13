/*
 * oracle/nhp_oracle.c -- TEST INFRASTRUCTURE (checker), not product code.
 * See nhp_oracle.h for scope, conventions and how parity is pinned.
 *
 * All reference citations are file:line under the reference checkout
 * (cswaney/NetworkHawkesProcesses.jl v0.1.0).  [3P] marks arithmetic that lives
 * in a pinned third-party package (Manifest.toml): Distributions 0.25.76,
 * StatsFuns 1.1.1, LogExpFunctions 0.3.19, SpecialFunctions 2.2.0, DSP 0.7.7,
 * Julia Base 1.8 reductions -- restated from their published algorithms.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include "nhp_oracle.h"
#include "nhp_detmath.h"

#include <stdlib.h>

#define IDX(p, c, N) ((size_t)(p) + (size_t)(c) * (size_t)(N))
static const double INVSQRT2PI = 0.3989422804014327; /* StatsFuns invsqrt2π [3P] */

static inline double xexp(double x, int flags) { return (flags & ORC_MATH_DET) ? nhp_det_exp(x) : exp(x); }
static inline double xlog(double x, int flags) { return (flags & ORC_MATH_DET) ? nhp_det_log(x) : log(x); }

double orc_det_exp(double x) { return nhp_det_exp(x); }
double orc_det_log(double x) { return nhp_det_log(x); }

/* ------------------------------------------------------------------ evaluators */

/* src/impulses.jl:106-108: pdf(Exponential(1 / θ[p,c]), Δt).
 * [3P] Distributions exponential.jl: λ = inv(scale); λ*exp(-λ*max(x,0)), 0 for x<0. */
double orc_impulse_exponential(double theta, double dt, int flags)
{
    double scale = 1.0 / theta;
    double r = 1.0 / scale;
    if (dt < 0.0) return 0.0;
    return r * xexp(-(r * dt), flags);
}

/* src/impulses.jl:174-178: pdf(LogitNormal(μ, τ^(-1/2)), Δt / Δtmax); note: NOT divided
 * by Δtmax.  [3P] Distributions logitnormal.jl + StatsFuns normpdf:
 * 0<x<1 ? exp(-z^2/2)*invsqrt2π/σ / (x*(1-x)) : 0,  z = (logit(x)-μ)/σ, logit = log(x/(1-x)).
 * The det variant is the cheaper fixed sequence the HIP kernels restate. */
double orc_impulse_logitnormal(double mu, double tau, double dt_max, double dt, int flags)
{
    if (flags & ORC_MATH_DET) {
        double inv = 1.0 / dt_max;
        double x = dt * inv;
        if (!(x > 0.0 && x < 1.0)) return 0.0;
        double st = sqrt(tau);
        double o = 1.0 - x;
        double q = 1.0 / (x * o);
        double lx = nhp_det_log((x * x) * q);
        double z = (lx - mu) * st;
        double e = nhp_det_exp(-0.5 * (z * z));
        return (e * (INVSQRT2PI * st)) * q;
    }
    double x = dt / dt_max;
    if (!(x > 0.0 && x < 1.0)) return 0.0;
    double sigma = pow(tau, -0.5);
    double lx = log(x / (1.0 - x));
    double z = (lx - mu) / sigma;
    return (exp(-(z * z) / 2.0) * INVSQRT2PI / sigma) / (x * (1.0 - x));
}

/* src/utils/interpolation.jl:27-36.  Linear search in the reference; the bin found is the
 * same one a binary search finds (x strictly increasing). */
int orc_linear_interpolate(const double *x, const double *y, int32_t n, double x0, double *out)
{
    if (x0 < x[0] || x0 > x[n - 1]) return ORC_EDOMAIN;
    for (int32_t i = 0; i + 1 < n; ++i) {
        if (x0 >= x[i] && x0 < x[i + 1]) {
            *out = (y[i + 1] * (x0 - x[i]) + y[i] * (x[i + 1] - x0)) / (x[i + 1] - x[i]);
            return ORC_OK;
        }
    }
    *out = y[n - 1];
    return ORC_OK;
}

/* src/utils/interpolation.jl:40-48 (trapezoid rule). */
double orc_linear_integrate(const double *x, const double *y, int32_t n)
{
    double I = 0.0;
    for (int32_t i = 0; i + 1 < n; ++i) I += 0.5 * (y[i] + y[i + 1]) * (x[i + 1] - x[i]);
    return I;
}

/* src/baselines.jl:115-118 (homogeneous; DomainError for time<0), :332-334 (LGCP). */
int orc_baseline_intensity(const orc_cont_model *m, int64_t node1, double t, double *out)
{
    if (node1 < 1 || node1 > m->n_nodes) return ORC_EDOMAIN;
    if (m->baseline_kind == ORC_BASELINE_HOMOGENEOUS) {
        if (t < 0.0) return ORC_EDOMAIN;
        *out = m->lambda0[node1 - 1];
        return ORC_OK;
    }
    return orc_linear_interpolate(m->grid_x, m->lambda0 + (size_t)(node1 - 1) * m->grid_n,
                                  m->grid_n, t, out);
}

/* src/baselines.jl:98-102 (λ .* duration), :336 (trapezoid, ignores duration). */
int orc_baseline_integral(const orc_cont_model *m, double duration, double *out)
{
    if (m->baseline_kind == ORC_BASELINE_HOMOGENEOUS) {
        if (duration < 0.0) return ORC_EDOMAIN;
        for (int32_t c = 0; c < m->n_nodes; ++c) out[c] = m->lambda0[c] * duration;
    } else {
        for (int32_t c = 0; c < m->n_nodes; ++c)
            out[c] = orc_linear_integrate(m->grid_x, m->lambda0 + (size_t)c * m->grid_n, m->grid_n);
    }
    return ORC_OK;
}

/* impulse_response(process, p, c, Δt): src/continuous.jl:302-305 (w * pdf) and
 * :521-525 (a * w * pdf, evaluated left to right).  p, c are 0-based here. */
static inline double pair_weight(const orc_cont_model *m, int32_t p, int32_t c, double dt, int flags)
{
    size_t k = IDX(p, c, m->n_nodes);
    double pdf = (m->impulse_kind == ORC_IMPULSE_EXPONENTIAL)
                     ? orc_impulse_exponential(m->theta[k], dt, flags)
                     : orc_impulse_logitnormal(m->mu[k], m->tau[k], m->dt_max, dt, flags);
    double w = m->W[k];
    if (m->A) return (m->A[k] * w) * pdf;
    return w * pdf;
}

static int validate_data(const orc_cont_model *m, const double *times, const int64_t *nodes, int64_t M)
{
    for (int64_t i = 0; i < M; ++i) {
        if (nodes[i] < 1 || nodes[i] > m->n_nodes) return ORC_EDOMAIN;
        if (!(times[i] >= 0.0)) return ORC_EDOMAIN;
        if (i > 0 && times[i] < times[i - 1]) return ORC_EINVAL;
    }
    return ORC_OK;
}

/* total_intensity: src/continuous.jl:286-300 (standard), :391-405 (network).
 * i is 0-based; the walk is most-recent-first and stops at the first parent that fails
 * events[j] > time - Δtmax (strict). */
static int total_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                           int64_t i, int flags, double *out)
{
    double t = times[i];
    int32_t c = (int32_t)(nodes[i] - 1);
    double lam;
    int rc = orc_baseline_intensity(m, nodes[i], t, &lam);
    if (rc) return rc;
    if (i > 0) {
        double thr = t - m->dt_max;
        for (int64_t j = i - 1; j >= 0 && times[j] > thr; --j)
            lam += pair_weight(m, (int32_t)(nodes[j] - 1), c, t - times[j], flags);
    }
    *out = lam;
    return ORC_OK;
}

int orc_cont_total_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                             int64_t M, int64_t i0, int64_t i1, int flags, double *lambda)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    for (int64_t i = i0; i < i1; ++i) {
        rc = total_intensity(m, times, nodes, i, flags, &lambda[i - i0]);
        if (rc) return rc;
    }
    return ORC_OK;
}

int64_t orc_cont_pair_count(const double *times, int64_t M, double dt_max)
{
    int64_t pairs = 0, first = 0;
    for (int64_t i = 0; i < M; ++i) {
        double thr = times[i] - dt_max;
        while (first < i && !(times[first] > thr)) ++first;
        pairs += i - first;
    }
    return pairs;
}

/* The -∫λ terms shared by both formulations: src/continuous.jl:216-221 (standard),
 * :367-371 (network, masked) and :245-248 / :411-414 (recursive; the network twin does NOT
 * mask with A -- SURVEY D7 -- restated literally via `masked`). */
static int integral_terms(const orc_cont_model *m, const int64_t *nodes, int64_t M, double duration,
                          int masked, int flags, double *ll)
{
    int32_t N = m->n_nodes;
    double *I0 = (double *)malloc(sizeof(double) * (size_t)N);
    int rc = orc_baseline_integral(m, duration, I0);
    if (rc) { free(I0); return rc; }
    double s = 0.0;
    for (int32_t c = 0; c < N; ++c) s += I0[c];
    free(I0);
    double acc = 0.0;
    acc -= s;
    const double *A = masked ? m->A : NULL;
    if (flags & ORC_FAST_INTEGRAL) {
        double *rows = (double *)calloc((size_t)N, sizeof(double));
        for (int32_t c = 0; c < N; ++c)
            for (int32_t p = 0; p < N; ++p)
                rows[p] += A ? A[IDX(p, c, N)] * m->W[IDX(p, c, N)] : m->W[IDX(p, c, N)];
        for (int64_t i = 0; i < M; ++i) acc -= rows[nodes[i] - 1];
        free(rows);
    } else {
        for (int64_t i = 0; i < M; ++i) {
            int32_t p = (int32_t)(nodes[i] - 1);
            double r = 0.0;
            for (int32_t c = 0; c < N; ++c)
                r += A ? A[IDX(p, c, N)] * m->W[IDX(p, c, N)] : m->W[IDX(p, c, N)];
            acc -= r;
        }
    }
    *ll = acc;
    return ORC_OK;
}

/* loglikelihood(...; recursive=false): src/continuous.jl:210-239 (serial branch :233-237),
 * network twin :360-389. */
int orc_cont_loglik_windowed(const orc_cont_model *m, const double *times, const int64_t *nodes,
                             int64_t M, double duration, int flags, double *ll_out)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    double ll;
    rc = integral_terms(m, nodes, M, duration, 1, flags, &ll);
    if (rc) return rc;
    for (int64_t i = 0; i < M; ++i) {
        double lam;
        rc = total_intensity(m, times, nodes, i, flags, &lam);
        if (rc) return rc;
        ll += xlog(lam, flags);
    }
    *ll_out = ll;
    return ORC_OK;
}

/* The threaded branch of the same function: src/continuous.jl:224-232 (`Threads.@threads` over the events, one
 * atomic add per event), network twin :374-382.  OpenMP threads with a per-thread partial sum instead of the atomic --
 * the sum's association differs from the serial branch's exactly as the reference's own does.  Used as the all-cores CPU
 * baseline of bench.py (SURVEY 8d ii); `threads` <= 0 takes the OpenMP default. */
int orc_cont_loglik_windowed_mt(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                int64_t M, double duration, int flags, int threads, double *ll_out)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    double ll;
    rc = integral_terms(m, nodes, M, duration, 1, flags, &ll);
    if (rc) return rc;
    double acc = 0.0;
    int bad = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) reduction(+ : acc) reduction(| : bad) schedule(static)
#endif
    for (int64_t i = 0; i < M; ++i) {
        double lam;
        int r = total_intensity(m, times, nodes, i, flags, &lam);
        if (r) { bad |= r; continue; }
        acc += xlog(lam, flags);
    }
    (void)threads;
    if (bad) return bad;
    *ll_out = ll + acc;
    return ORC_OK;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* recursive_loglikelihood: src/continuous.jl:241-276 (standard), :407-442 (network;
 * effective_weight = a*w :527-531).  Ignores Δtmax (D8); parenttimes > 0.0 doubles as the
 * "node seen" flag (D9).  Exponential impulses only (:212). */
int orc_cont_loglik_recursive(const orc_cont_model *m, const double *times, const int64_t *nodes,
                              int64_t M, double duration, int flags, double *ll_out)
{
    if (m->impulse_kind != ORC_IMPULSE_EXPONENTIAL) return ORC_EINVAL;
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    int32_t N = m->n_nodes;
    double ll;
    rc = integral_terms(m, nodes, M, duration, 0, flags, &ll);
    if (rc) return rc;
    double *P = (double *)calloc((size_t)N * N, sizeof(double));
    double *ptime = (double *)calloc((size_t)N, sizeof(double));
    for (int64_t i = 0; i < M; ++i) {
        double t = times[i];
        int32_t c = (int32_t)(nodes[i] - 1);
        double lam;
        rc = orc_baseline_intensity(m, nodes[i], t, &lam);
        if (rc) { free(P); free(ptime); return rc; }
        if (ptime[c] > 0.0) {
            double dt = t - ptime[c];
            for (int32_t k = 0; k < N; ++k) {
                double next = xexp(-dt * m->theta[IDX(c, k, N)], flags);
                P[IDX(c, k, N)] = next * (1.0 + P[IDX(c, k, N)]);
            }
        }
        for (int32_t p = 0; p < N; ++p) {
            double pt = ptime[p];
            if (pt > 0.0) {
                double r;
                if (p == c) {
                    r = P[IDX(p, c, N)];
                } else {
                    double dt = t - pt;
                    double next = xexp(-dt * m->theta[IDX(p, c, N)], flags);
                    r = next * (1.0 + P[IDX(p, c, N)]);
                }
                double w = m->A ? m->A[IDX(p, c, N)] * m->W[IDX(p, c, N)] : m->W[IDX(p, c, N)];
                lam += w * m->theta[IDX(p, c, N)] * r;
            }
        }
        ll += xlog(lam, flags);
        ptime[c] = t;
    }
    free(P);
    free(ptime);
    *ll_out = ll;
    return ORC_OK;
}

/* intensity(process, data, time): src/continuous.jl:84-96; mask is strict on both sides
 * (time - Δtmax < events < time); result Q x N column-major like the reference's λs. */
int orc_cont_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                       int64_t M, const double *q, int64_t Q, int flags, double *out)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    int32_t N = m->n_nodes;
    for (int64_t k = 0; k < Q; ++k) {
        double t = q[k];
        double lo = t - m->dt_max;
        for (int32_t c = 0; c < N; ++c) {
            double lam = 0.0;
            for (int64_t j = 0; j < M; ++j) {
                if (lo < times[j] && times[j] < t)
                    lam += pair_weight(m, (int32_t)(nodes[j] - 1), c, t - times[j], flags);
                else if (!(times[j] < t))
                    break;
            }
            double b;
            rc = orc_baseline_intensity(m, c + 1, t, &b);
            if (rc) return rc;
            out[(size_t)k + (size_t)c * (size_t)Q] = b + lam;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ parent sampler */

/* [3P] Julia Base sum over a Vector{Any} (reduce.jl): sequential for n <= 1024 (the @simd
 * loop cannot vectorise boxed elements), otherwise split at ifirst + (ilast-ifirst)>>1. */
static double julia_sum(const double *a, int64_t ifirst, int64_t ilast)
{
    if (ifirst == ilast) return a[ifirst];
    if (ilast - ifirst < 1024) {
        double v = a[ifirst] + a[ifirst + 1];
        for (int64_t i = ifirst + 2; i <= ilast; ++i) v += a[i];
        return v;
    }
    int64_t imid = ifirst + ((ilast - ifirst) >> 1);
    return julia_sum(a, ifirst, imid) + julia_sum(a, imid + 1, ilast);
}

/* resample_parents / resample_parent: src/parents.jl:1-46.  Weights most-recent-first,
 * baseline last (:32-41); p = λs ./ sum(λs); [3P] Distributions rand(DiscreteNonParametric):
 * cp = p[1]; i = 1; while cp <= u && i < n: cp += p[i += 1].  u[i] is the explicit uniform
 * for event i (the reference uses the task-local RNG, which is unreproducible under
 * threads; the explicit stream is the contract -- SURVEY 7, hard part 4). */
int orc_cont_resample_parents(const orc_cont_model *m, const double *times, const int64_t *nodes,
                              int64_t M, const double *u, int flags,
                              int64_t *parents, int64_t *parentnodes)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    int64_t cap = 1024;
    double *w = (double *)malloc(sizeof(double) * (size_t)cap);
    for (int64_t i = 0; i < M; ++i) {
        if (i == 0) { parents[0] = 0; parentnodes[0] = 0; continue; }
        double t = times[i];
        int32_t c = (int32_t)(nodes[i] - 1);
        double thr = t - m->dt_max;
        int64_t n = 0;
        for (int64_t j = i - 1; j >= 0 && times[j] > thr; --j) {
            if (n + 2 > cap) { cap *= 2; w = (double *)realloc(w, sizeof(double) * (size_t)cap); }
            w[n++] = pair_weight(m, (int32_t)(nodes[j] - 1), c, t - times[j], flags);
        }
        double b;
        rc = orc_baseline_intensity(m, nodes[i], t, &b);
        if (rc) { free(w); return rc; }
        w[n++] = b;
        double s = julia_sum(w, 0, n - 1);
        if (!(s > 0.0) || !(s < INFINITY)) { free(w); return ORC_EDOMAIN; }
        double draw = u[i];
        int64_t k = 0;
        double cp = w[0] / s;
        while (cp <= draw && k < n - 1) { ++k; cp += w[k] / s; }
        if (k == n - 1) { parents[i] = 0; parentnodes[i] = 0; }
        else { parents[i] = i - k; /* 1-based index of event (i-1-k) */ parentnodes[i] = nodes[i - 1 - k]; }
    }
    free(w);
    return ORC_OK;
}

void orc_uniform_stream(uint64_t seed, uint64_t step, int64_t M, double *u)
{
    for (int64_t i = 0; i < M; ++i) u[i] = nhp_uniform(seed, step, (uint64_t)i);
}

/* node_counts: src/parents.jl:61-68 */
void orc_node_counts(const int64_t *nodes, int64_t M, int32_t N, double *Mn)
{
    for (int32_t c = 0; c < N; ++c) Mn[c] = 0.0;
    for (int64_t i = 0; i < M; ++i) Mn[nodes[i] - 1] += 1.0;
}

/* parent_counts: src/parents.jl:70-79 */
void orc_parent_counts(const int64_t *nodes, const int64_t *parentnodes, int64_t M, int32_t N, double *Mnm)
{
    for (size_t k = 0; k < (size_t)N * N; ++k) Mnm[k] = 0.0;
    for (int64_t i = 0; i < M; ++i)
        if (parentnodes[i] > 0) Mnm[IDX(parentnodes[i] - 1, nodes[i] - 1, N)] += 1.0;
}

/* node_counts(nodes, parentnodes, nnodes): src/baselines.jl:87-96 (pinned by
 * test/baselines.jl:8-25) */
void orc_baseline_node_counts(const int64_t *nodes, const int64_t *parentnodes, int64_t M, int32_t N, double *cnt0)
{
    for (int32_t c = 0; c < N; ++c) cnt0[c] = 0.0;
    for (int64_t i = 0; i < M; ++i)
        if (parentnodes[i] == 0) cnt0[nodes[i] - 1] += 1.0;
}

/* duration_mean: src/impulses.jl:84-96; fillna!(Xnm ./ Mnm, 0): src/utils/helpers.jl:18-25 */
void orc_duration_mean(const double *times, const int64_t *nodes, const int64_t *parents,
                       int64_t M, int32_t N, double *Xnm)
{
    double *Mnm = (double *)calloc((size_t)N * N, sizeof(double));
    for (size_t k = 0; k < (size_t)N * N; ++k) Xnm[k] = 0.0;
    for (int64_t i = 0; i < M; ++i) {
        int64_t par = parents[i];
        if (par > 0) {
            size_t k = IDX(nodes[par - 1] - 1, nodes[i] - 1, N);
            Mnm[k] += 1.0;
            Xnm[k] += times[i] - times[par - 1];
        }
    }
    for (size_t k = 0; k < (size_t)N * N; ++k) {
        double v = Xnm[k] / Mnm[k];
        Xnm[k] = (v != v) ? 0.0 : v;
    }
    free(Mnm);
}

/* log_duration_sum ./ Mnm and log_duration_variation: src/impulses.jl:216-252
 * (log_duration :228).  Xnm keeps NaN where Mnm == 0, as the reference does. */
void orc_log_duration_stats(const double *times, const int64_t *nodes, const int64_t *parents,
                            int64_t M, int32_t N, double dt_max, double *Xnm, double *Vnm)
{
    double *Mnm = (double *)calloc((size_t)N * N, sizeof(double));
    for (size_t k = 0; k < (size_t)N * N; ++k) { Xnm[k] = 0.0; Vnm[k] = 0.0; }
    for (int64_t i = 0; i < M; ++i) {
        int64_t par = parents[i];
        if (par > 0) {
            size_t k = IDX(nodes[par - 1] - 1, nodes[i] - 1, N);
            double d = times[i] - times[par - 1];
            Mnm[k] += 1.0;
            Xnm[k] += log(d / (dt_max - d));
        }
    }
    for (size_t k = 0; k < (size_t)N * N; ++k) Xnm[k] = Xnm[k] / Mnm[k];
    for (int64_t i = 0; i < M; ++i) {
        int64_t par = parents[i];
        if (par > 0) {
            size_t k = IDX(nodes[par - 1] - 1, nodes[i] - 1, N);
            double d = times[i] - times[par - 1];
            double e = log(d / (dt_max - d)) - Xnm[k];
            Vnm[k] += e * e;
        }
    }
    free(Mnm);
}

/* ------------------------------------------------------------------ adjacency Gibbs
 * resample_adjacency_matrix! / resample_column!: src/continuous.jl:444-487.
 * integrated_intensity(process, node, nodecounts, duration): :489-498.
 * sum_log_intensity: :500-519 -- note `index == 1 && continue` sits AFTER λ0 is read, so the very
 * first event's log λ0 is dropped from the sum (cancels between ll0 and ll1). */
static double adj_sum_log_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                    int64_t M, int32_t node0)
{
    double S = 0.0;
    for (int64_t i = 0; i < M; ++i) {
        if (nodes[i] - 1 != node0) continue;
        if (i == 0) continue;
        double lam;
        if (total_intensity(m, times, nodes, i, 0, &lam)) return NAN;
        S += log(lam);
    }
    return S;
}

static double adj_integrated_intensity(const orc_cont_model *m, int32_t node0, const double *cnt, double duration)
{
    double I;
    if (m->baseline_kind == ORC_BASELINE_HOMOGENEOUS) I = m->lambda0[node0] * duration;
    else I = orc_linear_integrate(m->grid_x, m->lambda0 + (size_t)node0 * m->grid_n, m->grid_n);
    for (int32_t p = 0; p < m->n_nodes; ++p)
        I += m->A[IDX(p, node0, m->n_nodes)] * m->W[IDX(p, node0, m->n_nodes)] * cnt[p];
    return I;
}

/* Columns [c0, c1) only: resample_column! (src/continuous.jl:466-487) reads and writes column c of the matrix
 * alone (integrated_intensity :489-498 and sum_log_intensity :500-519 are per child node), so a subset of the
 * columns gets the values the whole sweep gives it -- what the full-size parity tests use. */
int orc_cont_resample_adjacency_columns(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                        int64_t M, double duration, const double *rho, const double *u, double *A,
                                        int32_t c0, int32_t c1)
{
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    int32_t N = m->n_nodes;
    if (c0 < 0 || c1 > N || c0 > c1) return ORC_EINVAL;
    orc_cont_model w = *m;
    w.A = A;                                             /* work on the caller's matrix in place */
    double *cnt = (double *)malloc(sizeof(double) * (size_t)N);
    orc_node_counts(nodes, M, N, cnt);
    for (int32_t c = c0; c < c1; ++c)
        for (int32_t p = 0; p < N; ++p) {
            size_t k = IDX(p, c, N);
            A[k] = 0.0;
            double ll0 = -adj_integrated_intensity(&w, c, cnt, duration);
            ll0 += adj_sum_log_intensity(&w, times, nodes, M, c);
            ll0 += log(1.0 - rho[k]);
            A[k] = 1.0;
            double ll1 = -adj_integrated_intensity(&w, c, cnt, duration);
            ll1 += adj_sum_log_intensity(&w, times, nodes, M, c);
            ll1 += log(rho[k]);
            double mx = ll0 > ll1 ? ll0 : ll1;
            double Z = mx + log(exp(ll0 - mx) + exp(ll1 - mx));
            A[k] = (u[k] <= exp(ll1 - Z)) ? 1.0 : 0.0;
        }
    free(cnt);
    return ORC_OK;
}

int orc_cont_resample_adjacency(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                int64_t M, double duration, const double *rho, const double *u, double *A)
{
    return orc_cont_resample_adjacency_columns(m, times, nodes, M, duration, rho, u, A, 0, m->n_nodes);
}

/* ------------------------------------------------------------------ analytic gradient
 * No reference code: the reference hands Optim no gradient (src/continuous.jl:190), so one
 * gradient costs it 2P objective calls.  Formulas from SURVEY.md 7; validated in tests/ by
 * central finite differences of the ll functions above.  Output order is params! order
 * (src/continuous.jl:121-129): [λ0 (N); θ (N²) | μ (N²); τ (N²); W (N²)].  Homogeneous
 * baseline only.  recursive != 0 differentiates the recursive formulation (all earlier
 * parents with time > 0, unmasked integral). */
int orc_cont_loglik_grad(const orc_cont_model *m, const double *times, const int64_t *nodes,
                         int64_t M, double duration, int recursive, double *ll_out, double *grad)
{
    if (recursive && m->impulse_kind != ORC_IMPULSE_EXPONENTIAL) return ORC_EINVAL;
    int rc = validate_data(m, times, nodes, M);
    if (rc) return rc;
    int32_t N = m->n_nodes;
    size_t NN = (size_t)N * N;
    int lognorm = m->impulse_kind == ORC_IMPULSE_LOGITNORMAL;
    const int lgcp = m->baseline_kind != ORC_BASELINE_HOMOGENEOUS;
    const int32_t G = m->grid_n;
    const size_t nb = lgcp ? (size_t)N * G : (size_t)N;            /* params(baseline): λ or vcat(λ...) */
    size_t P = nb + NN * (lognorm ? 3 : 2);
    for (size_t k = 0; k < P; ++k) grad[k] = 0.0;
    double *g0 = grad, *g1 = grad + nb, *g2 = lognorm ? grad + nb + NN : NULL;
    double *gW = grad + nb + NN * (lognorm ? 2 : 1);
    double ll;
    rc = integral_terms(m, nodes, M, duration, recursive ? 0 : 1, 0, &ll);
    if (rc) return rc;
    if (!lgcp) {
        for (int32_t c = 0; c < N; ++c) g0[c] = -duration;
    } else {                                                       /* d/dy_g of the trapezoid rule */
        const double *x = m->grid_x;
        for (int32_t c = 0; c < N; ++c)
            for (int32_t g = 0; g < G; ++g) {
                double left = g > 0 ? x[g] - x[g - 1] : 0.0, right = g + 1 < G ? x[g + 1] - x[g] : 0.0;
                g0[(size_t)c * G + g] = -0.5 * (left + right);
            }
    }
    for (int64_t i = 0; i < M; ++i) {
        int32_t p = (int32_t)(nodes[i] - 1);
        for (int32_t c = 0; c < N; ++c)
            gW[IDX(p, c, N)] -= (m->A && !recursive) ? m->A[IDX(p, c, N)] : 1.0;
    }
    double *dW = (double *)malloc(sizeof(double) * (size_t)N);
    double *d1 = (double *)malloc(sizeof(double) * (size_t)N);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)N);
    for (int64_t i = 0; i < M; ++i) {
        double t = times[i];
        int32_t c = (int32_t)(nodes[i] - 1);
        double lam;
        rc = orc_baseline_intensity(m, c + 1, t, &lam);
        if (rc) { free(dW); free(d1); free(d2); return rc; }
        for (int32_t p = 0; p < N; ++p) { dW[p] = 0.0; d1[p] = 0.0; d2[p] = 0.0; }
        double thr = recursive ? -INFINITY : t - m->dt_max;
        for (int64_t j = i - 1; j >= 0 && times[j] > thr; --j) {
            if (recursive && !(times[j] > 0.0)) continue;
            int32_t p = (int32_t)(nodes[j] - 1);
            size_t k = IDX(p, c, N);
            double a = m->A ? m->A[k] : 1.0, w = m->W[k], dt = t - times[j];
            if (!lognorm) {
                double th = m->theta[k], e = exp(-th * dt);
                lam += a * w * th * e;
                dW[p] += a * th * e;
                d1[p] += a * w * (1.0 - th * dt) * e;
            } else {
                double x = dt / m->dt_max;
                if (!(x > 0.0 && x < 1.0)) continue;
                double tau = m->tau[k], mu = m->mu[k];
                double l = log(x / (1.0 - x)), dlt = l - mu;
                double h = exp(-0.5 * tau * dlt * dlt) * sqrt(tau) * INVSQRT2PI / (x * (1.0 - x));
                lam += a * w * h;
                dW[p] += a * h;
                d1[p] += a * w * h * tau * dlt;
                d2[p] += a * w * h * (0.5 / tau - 0.5 * dlt * dlt);
            }
        }
        ll += log(lam);
        double g = 1.0 / lam;
        if (!lgcp) {
            g0[c] += g;
        } else {                                                   /* interpolation weights of the two grid neighbours */
            const double *x = m->grid_x;
            int32_t lo = G - 1;
            for (int32_t q = 0; q + 1 < G; ++q) if (t >= x[q] && t < x[q + 1]) { lo = q; break; }
            if (lo == G - 1) {
                g0[(size_t)c * G + G - 1] += g;
            } else {
                g0[(size_t)c * G + lo] += g * (x[lo + 1] - t) / (x[lo + 1] - x[lo]);
                g0[(size_t)c * G + lo + 1] += g * (t - x[lo]) / (x[lo + 1] - x[lo]);
            }
        }
        for (int32_t p = 0; p < N; ++p) {
            size_t k = IDX(p, c, N);
            gW[k] += g * dW[p];
            g1[k] += g * d1[p];
            if (lognorm) g2[k] += g * d2[p];
        }
    }
    free(dW); free(d1); free(d2);
    *ll_out = ll;
    return ORC_OK;
}

/* ------------------------------------------------------------------ discrete path */

/* basis: src/impulses.jl:321-335.  σ = L/(B-1); means: interior points of
 * LinRange(1,L,B+2) if B < L else LinRange(1,L,B) ([3P] Base LinRange: (1-t)*a + t*b,
 * t = (i-1)/(len-1)); ϕ = exp.(-1 / 2 * σ^-1 / 2 .* d.^2) which Julia parses as
 * (((-1/2) * inv(σ)) / 2) * d²; normalised by sum(ϕ_b) * dt. */
int orc_disc_basis(int32_t L, int32_t B, double dt, double *phi)
{
    if (L < 1 || B < 1) return ORC_EINVAL;
    double sigma = (double)L / (double)(B - 1);
    double coef = ((-1.0 / 2.0) * (1.0 / sigma)) / 2.0;
    for (int32_t b = 0; b < B; ++b) {
        int32_t len = (B < L) ? B + 2 : B;
        int32_t i = (B < L) ? b + 1 : b;
        double tt = (double)i / (double)(len > 1 ? len - 1 : 1);
        double mu = (1.0 - tt) * 1.0 + tt * (double)L;
        double s = 0.0;
        for (int32_t l = 0; l < L; ++l) {
            double d = (double)(l + 1) - mu;
            phi[l + (size_t)b * L] = exp(coef * (d * d));
            s += phi[l + (size_t)b * L];
        }
        for (int32_t l = 0; l < L; ++l) phi[l + (size_t)b * L] /= (s * dt);
    }
    return ORC_OK;
}

/* convolve: src/discrete.jl:146-151.  conv(transpose(data), [0.0; ϕ_b])[1:T, :] is
 * Ŝ[t,n,b] = Σ_{l=1..min(L,t-1)} data[n,t-l] ϕ_b[l] (1-based t); the reference's DSP.conv
 * [3P] is FFT-based, so its output carries ~1e-16 noise that max.(·, 0) clips; the direct
 * sum here is the exact value of the same quantity.  data is N x T column-major. */
void orc_disc_convolve(const int64_t *data, int32_t N, int64_t T, const double *phi, int32_t L,
                       int32_t B, double *conv)
{
    for (int32_t b = 0; b < B; ++b)
        for (int32_t n = 0; n < N; ++n)
            for (int64_t t = 0; t < T; ++t) {
                double s = 0.0;
                int64_t lmax = t < L ? t : L;
                for (int64_t l = 1; l <= lmax; ++l)
                    s += (double)data[n + (size_t)(t - l) * N] * phi[(l - 1) + (size_t)b * L];
                conv[(size_t)t + (size_t)n * T + (size_t)b * T * N] = s > 0.0 ? s : 0.0;
            }
}

/* intensity(process, convolved): src/discrete.jl:115-129; bump :381-385 (w*θ*dt) and
 * :511-516 (a*w*θ*dt); baseline src/baselines.jl:402-405 (λ .* dt). */
void orc_disc_intensity_b(const double *conv, int64_t T, int32_t N, int32_t B, const double *lambda0,
                          const double *base_tn, const double *W, const double *theta, const double *A, double dt,
                          double *lam)
{
    for (int64_t t = 0; t < T; ++t)
        for (int32_t c = 0; c < N; ++c) {
            double v = base_tn ? base_tn[(size_t)t + (size_t)c * T] : lambda0[c] * dt;
            for (int32_t p = 0; p < N; ++p)
                for (int32_t b = 0; b < B; ++b) {
                    double shat = conv[(size_t)t + (size_t)p * T + (size_t)b * T * N];
                    double w = W[IDX(p, c, N)];
                    double th = theta[IDX(p, c, N) + (size_t)b * N * N];
                    double bump = A ? A[IDX(p, c, N)] * w * th * dt : w * th * dt;
                    v += shat * bump;
                }
            lam[(size_t)t + (size_t)c * T] = v;
        }
}

void orc_disc_intensity(const double *conv, int64_t T, int32_t N, int32_t B, const double *lambda0,
                        const double *W, const double *theta, const double *A, double dt, double *lam)
{
    orc_disc_intensity_b(conv, T, N, B, lambda0, NULL, W, theta, A, dt, lam);
}

/* intensity(p::DiscreteLogGaussianCoxProcess, times): src/baselines.jl:531-537 -- column n is the linear
 * interpolation of (x, λ[:, n] * dt); lam is G x N column-major; out[i + n*ntimes]. */
int orc_disc_lgcp_intensity(const double *x, int32_t G, const double *lam, int32_t N, double dt,
                            const double *times, int64_t ntimes, double *out)
{
    double *y = (double *)malloc(sizeof(double) * (size_t)G);
    if (!y) return ORC_ENOMEM;
    for (int32_t n = 0; n < N; ++n) {
        for (int32_t g = 0; g < G; ++g) y[g] = lam[(size_t)g + (size_t)n * G] * dt;
        for (int64_t i = 0; i < ntimes; ++i) {
            int rc = orc_linear_interpolate(x, y, G, times[i], &out[(size_t)i + (size_t)n * ntimes]);
            if (rc) { free(y); return rc; }
        }
    }
    free(y);
    return ORC_OK;
}

/* loglikelihood(process, data, convolved): src/discrete.jl:91-102.
 * [3P] pdf(Poisson(λ), s) = exp(xlogy(s, λ) - λ - loggamma(s+1)); the reference then takes
 * log() of it (underflows to -Inf below e^-745), restated literally. */
double orc_disc_loglik(const int64_t *data, const double *lam, int64_t T, int32_t N)
{
    double ll = 0.0;
    for (int64_t t = 0; t < T; ++t)
        for (int32_t n = 0; n < N; ++n) {
            double s = (double)data[n + (size_t)t * N];
            double l = lam[(size_t)t + (size_t)n * T];
            double xlogy = (s == 0.0) ? 0.0 : s * log(l);
            double v = xlogy - l - lgamma(s + 1.0);
            ll += log(exp(v));
        }
    return ll;
}

/* [3P] SpecialFunctions.digamma for x > 0: recurrence up to x >= 10, then the asymptotic
 * series to x^-14 (truncation < 5e-17). */
double orc_digamma(double x)
{
    if (!(x > 0.0)) return NAN;
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    double f = 1.0 / (x * x);
    double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
               f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}

/* update!(process, data, convolved): src/discrete.jl:369-375.  update_parents
 * src/parents.jl:136-177 materialises u[T,N,1+NB] from the OLD variational parameters
 * (slot 1 = baseline, slot 1+(p-1)B+b = parent p, basis b), normalises over the last axis,
 * then: baseline src/baselines.jl:444-452 (βv = 1/β0 + T·dt), weights
 * src/weights.jl:70-91, impulses src/impulses.jl:355-371; log-expectations
 * src/baselines.jl:454-456, src/weights.jl:95-97, src/impulses.jl:373-375. */
int orc_disc_vb_step(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                     double dt, double alpha0, double beta0, double kappa, double nu, double gamma,
                     double *alpha_v, double *beta_v, double *kappa_v, double *nu_v, double *gamma_v)
{
    size_t NN = (size_t)N * N, K = 1 + (size_t)N * B;
    double *u = (double *)malloc(sizeof(double) * (size_t)T * N * K);
    if (!u) return ORC_EINVAL;
    double *e0 = (double *)malloc(sizeof(double) * (size_t)N);
    double *E = (double *)malloc(sizeof(double) * NN * B);
    for (int32_t c = 0; c < N; ++c) e0[c] = exp(orc_digamma(alpha_v[c]) - log(beta_v[c]));
    for (int32_t p = 0; p < N; ++p)
        for (int32_t c = 0; c < N; ++c) {
            double gs = 0.0;
            for (int32_t b = 0; b < B; ++b) gs += gamma_v[IDX(p, c, N) + (size_t)b * NN];
            double elw = orc_digamma(kappa_v[IDX(p, c, N)]) - log(nu_v[IDX(p, c, N)]);
            for (int32_t b = 0; b < B; ++b) {
                double elt = orc_digamma(gamma_v[IDX(p, c, N) + (size_t)b * NN]) - orc_digamma(gs);
                E[IDX(p, c, N) + (size_t)b * NN] = exp(elt + elw);
            }
        }
#define U(t, c, k) u[(size_t)(t) + (size_t)(c) * T + (size_t)(k) * T * N]
    for (int64_t t = 0; t < T; ++t)
        for (int32_t c = 0; c < N; ++c) {
            double Z = 0.0;
            U(t, c, 0) = e0[c];
            for (int32_t p = 0; p < N; ++p)
                for (int32_t b = 0; b < B; ++b)
                    U(t, c, 1 + (size_t)p * B + b) =
                        conv[(size_t)t + (size_t)p * T + (size_t)b * T * N] * E[IDX(p, c, N) + (size_t)b * NN];
            for (size_t k = 0; k < K; ++k) Z += U(t, c, k);
            for (size_t k = 0; k < K; ++k) U(t, c, k) /= Z;
        }
    for (int32_t c = 0; c < N; ++c) {
        double s = 0.0;
        for (int64_t t = 0; t < T; ++t) s += U(t, c, 0) * (double)data[c + (size_t)t * N];
        alpha_v[c] = alpha0 + s;
        beta_v[c] = 1.0 / beta0 + (double)T * dt;
    }
    for (int32_t p = 0; p < N; ++p)
        for (int32_t c = 0; c < N; ++c) {
            double k1 = 0.0, n1 = 0.0;
            for (int64_t t = 0; t < T; ++t) {
                double sp = (double)data[p + (size_t)t * N], sc = (double)data[c + (size_t)t * N];
                for (int32_t b = 0; b < B; ++b) k1 += sc * U(t, c, 1 + (size_t)p * B + b);
                n1 += sp;
            }
            kappa_v[IDX(p, c, N)] = kappa + k1;
            nu_v[IDX(p, c, N)] = nu + n1;
            for (int32_t b = 0; b < B; ++b) {
                double g = 0.0;
                for (int64_t t = 0; t < T; ++t)
                    g += (double)data[c + (size_t)t * N] * U(t, c, 1 + (size_t)p * B + b);
                gamma_v[IDX(p, c, N) + (size_t)b * NN] = gamma + g;
            }
        }
#undef U
    free(u); free(e0); free(E);
    return ORC_OK;
}

/* loglikelihood(process::LogGaussianCoxProcess, data, node, y) for every node at once:
 * src/baselines.jl:247-254 on the events split_extract (:227-238) keeps -- node c's events whose
 * sampled parent node is 0 (the baseline).  lam[c*G + k] = exp(m + y_c[k]) is the candidate
 * intensity on the grid; ll[c] = -trapezoid(lam_c) + sum_i log lam_c(t_i). */
int orc_lgcp_loglik(const double *times, const int64_t *nodes, const int64_t *parentnodes, int64_t M,
                    int32_t N, const double *grid_x, int32_t G, const double *lam, double *ll)
{
    for (int32_t c = 0; c < N; ++c) ll[c] = -orc_linear_integrate(grid_x, lam + (size_t)c * G, G);
    for (int64_t i = 0; i < M; ++i) {
        if (parentnodes[i] != 0) continue;
        const int64_t c = nodes[i] - 1;
        if (c < 0 || c >= N) return ORC_EDOMAIN;
        double f;
        int rc = orc_linear_interpolate(grid_x, lam + (size_t)c * G, G, times[i], &f);
        if (rc != ORC_OK) return rc;
        ll[c] += log(f);
    }
    return ORC_OK;
}

/* Discrete Gibbs parent counts: resample_parents / resample_parent src/parents.jl:82-116 reduced over
 * time, counts[c + N*k] = sum_t parents[t, c, k] (what every discrete resample! consumes:
 * src/baselines.jl:413-419, src/weights.jl:28-35, src/impulses.jl:337-353).  Bin (t, c) with
 * n = data[c, t] events draws Multinomial(n, mu), mu = [lambda0_c dt, shat[t,p,b] bump(p,c,b) ...]
 * normalised, categories in the reference's order k = 1 + p*B + b.
 *
 * [3P] Distributions' Multinomial sampler cannot be matched bit for bit (Julia RNG); the draw is
 * restated as n iid categorical draws through explicit uniforms -- the same distribution: the n
 * uniforms are produced in ascending order (order statistics, u_(j) = u_(j-1) + (1 - u_(j-1)) *
 * (1 - V_j^(1/(n-j))), V_j = Philox(seed ^ ORC_DISC_KEY, step, (bin << 20) | j), bin = t + T*c) and
 * each takes the smallest k whose cumulative sum exceeds u * total (capped at the last category),
 * the rule of the continuous sampler.  det-math exp/log, separate multiply and add. */
#define ORC_DISC_KEY 0xD15C0DE5EEDC0FFEull
static double disc_next_u(double u_prev, int64_t remaining, uint64_t seed, uint64_t step, uint64_t bin, int64_t j)
{
    double V = nhp_uniform(seed ^ ORC_DISC_KEY, step, (bin << 20) | (uint64_t)j);
    double r = nhp_det_exp(nhp_det_log(V) / (double)remaining);
    double w = 1.0 - r;
    return u_prev + (1.0 - u_prev) * w;
}

int orc_disc_resample_parents_b(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                                const double *lambda0, const double *base_tn, const double *W, const double *theta,
                                const double *A, double dt, uint64_t seed, uint64_t step, int64_t *counts,
                                int64_t *base_counts /* nullable: [T*N] parents[t, c, 1] */)
{
    const int64_t K = (int64_t)N * B;
    for (int64_t i = 0; i < (int64_t)N * (1 + K); ++i) counts[i] = 0;
    if (base_counts) for (int64_t i = 0; i < T * N; ++i) base_counts[i] = 0;
    double *bump = (double *)malloc(sizeof(double) * (size_t)K);
    if (!bump) return ORC_ENOMEM;
    for (int32_t c = 0; c < N; ++c) {
        for (int32_t p = 0; p < N; ++p)
            for (int32_t b = 0; b < B; ++b) {
                double w = W[IDX(p, c, N)], th = theta[IDX(p, c, N) + (size_t)b * N * N];
                bump[(size_t)p * B + b] = A ? A[IDX(p, c, N)] * w * th * dt : w * th * dt;
            }
        for (int64_t t = 0; t < T; ++t) {
            const int64_t n = data[c + (size_t)t * N];
            if (n <= 0) continue;
            const double base = base_tn ? base_tn[(size_t)t + (size_t)c * T] : lambda0[c] * dt;
            double total = base;
            for (int32_t p = 0; p < N; ++p)
                for (int32_t b = 0; b < B; ++b)
                    total = total + conv[(size_t)t + (size_t)p * T + (size_t)b * T * N] * bump[(size_t)p * B + b];
            const uint64_t bin = (uint64_t)t + (uint64_t)T * (uint64_t)c;
            int64_t j = 0;
            double u = disc_next_u(0.0, n, seed, step, bin, 0), thr = u * total;
            double cum = base;
            while (j < n && cum > thr) {
                counts[c] += 1;
                if (base_counts) base_counts[(size_t)t + (size_t)c * T] += 1;
                if (++j < n) { u = disc_next_u(u, n - j, seed, step, bin, j); thr = u * total; }
            }
            for (int32_t p = 0; p < N && j < n; ++p)
                for (int32_t b = 0; b < B && j < n; ++b) {
                    cum = cum + conv[(size_t)t + (size_t)p * T + (size_t)b * T * N] * bump[(size_t)p * B + b];
                    while (j < n && cum > thr) {
                        counts[c + (size_t)N * (1 + (size_t)p * B + b)] += 1;
                        if (++j < n) { u = disc_next_u(u, n - j, seed, step, bin, j); thr = u * total; }
                    }
                }
            counts[c + (size_t)N * K] += n - j;                  /* capped at the last category */
        }
    }
    free(bump);
    return ORC_OK;
}

int orc_disc_resample_parents(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                              const double *lambda0, const double *W, const double *theta, const double *A,
                              double dt, uint64_t seed, uint64_t step, int64_t *counts)
{
    return orc_disc_resample_parents_b(data, conv, T, N, B, lambda0, NULL, W, theta, A, dt, seed, step, counts, NULL);
}

/* loglikelihood(p::DiscreteLogGaussianCoxProcess, data, node, y) for every node: src/baselines.jl:571-584.
 * s0 [T*N] (t fastest) = the baseline-attributed counts; cand [G*N] = exp.(m .+ y) per node;
 * times = range(p) = x[1] : dt : x[end] - dt. */
int orc_disc_lgcp_loglik(const int64_t *s0, int64_t T, int32_t N, const double *x, int32_t G, const double *cand,
                         double dt, double *ll)
{
    double *y = (double *)malloc(sizeof(double) * (size_t)G);
    if (!y) return ORC_ENOMEM;
    for (int32_t n = 0; n < N; ++n) {
        for (int32_t g = 0; g < G; ++g) y[g] = cand[(size_t)g + (size_t)n * G] * dt;
        double acc = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            double lam;
            int rc = orc_linear_interpolate(x, y, G, x[0] + (double)t * dt, &lam);
            if (rc) { free(y); return rc; }
            double sd = (double)s0[(size_t)t + (size_t)n * T];
            double xlogy = (sd == 0.0) ? 0.0 : sd * log(lam);
            acc += log(exp(xlogy - lam - lgamma(sd + 1.0)));
        }
        ll[n] = acc;
    }
    free(y);
    return ORC_OK;
}

/* Discrete adjacency Gibbs: resample_adjacency_matrix! / resample_column! / conditional_loglikelihood
 * src/discrete.jl:424-480, restated literally -- for every entry two full passes over all T bins and all
 * N*B parent terms of the child column (O(N^3 B T) per sweep: small cases only).
 * [3P] pdf(Poisson(λ), s) = exp(xlogy(s, λ) - λ - loggamma(s+1)), then log(); rand(Bernoulli(q)) = u <= q;
 * logsumexp src/utils/helpers.jl:13-16.  rho[p + c*N] = link_probability, u[p + c*N] the explicit uniforms;
 * A is updated in place, column by column, parent by parent. */
static double disc_conditional_ll(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                                  const double *lambda0, const double *W, const double *theta, const double *A,
                                  double dt, double value, int32_t pidx, int32_t cidx)
{
    double ll = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        double lam = lambda0[cidx] * dt;                     /* intensity(process.baseline, 1:T)[t, cidx] */
        for (int32_t p = 0; p < N; ++p) {
            double w = W[IDX(p, cidx, N)];
            double a = (p == pidx) ? value : A[IDX(p, cidx, N)];
            for (int32_t b = 0; b < B; ++b) {
                double shat = conv[(size_t)t + (size_t)p * T + (size_t)b * T * N];
                double th = theta[IDX(p, cidx, N) + (size_t)b * N * N];
                lam += shat * a * w * th * dt;
            }
        }
        double sd = (double)data[cidx + (size_t)t * N];
        double xlogy = (sd == 0.0) ? 0.0 : sd * log(lam);
        ll += log(exp(xlogy - lam - lgamma(sd + 1.0)));
    }
    return ll;
}

int orc_disc_resample_adjacency(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                                const double *lambda0, const double *W, const double *theta, double dt,
                                const double *rho, const double *u, double *A)
{
    for (int32_t c = 0; c < N; ++c)
        for (int32_t p = 0; p < N; ++p) {
            size_t k = IDX(p, c, N);
            double ll0 = disc_conditional_ll(data, conv, T, N, B, lambda0, W, theta, A, dt, 0.0, p, c) + log(1.0 - rho[k]);
            double ll1 = disc_conditional_ll(data, conv, T, N, B, lambda0, W, theta, A, dt, 1.0, p, c) + log(rho[k]);
            double mx = ll0 > ll1 ? ll0 : ll1;
            double Z = mx + log(exp(ll0 - mx) + exp(ll1 - mx));
            A[k] = (u[k] <= exp(ll1 - Z)) ? 1.0 : 0.0;
        }
    return ORC_OK;
}

