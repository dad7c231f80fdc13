/*
 * oracle/nhp_oracle.h -- TEST INFRASTRUCTURE (checker), not product code.
 *
 * Plain-C, single-thread, fp64 restatement of the reference's event-history
 * intensity / log-likelihood hot path (cswaney/NetworkHawkesProcesses.jl v0.1.0).
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libnhp.so) never does.
 *
 * PARITY PINNING: the reference's own tests hold no golden vector for this path
 *  and Julia is absent from the build container, so the oracle is
 * pinned by (1) the few reference fixtures that touch it (test/baselines.jl:8-25,
 * 69-81; test/interpolation.jl:7-14), (2) closed-form known answers, (3) a
 * 50-digit mpmath evaluator, (4) agreement of the two independent formulations
 * (windowed at dt_max=Inf == recursive) and (5) scipy cross-checks of the
 * third-party pieces.  Results no reference fixture covers are "parity unpinned"
 * against the Julia package itself.
 *
 * Conventions: Julia layout.  Matrices are column-major, X[p,c] at p + c*N with
 * p = parent, c = child (src/continuous.jl:303).  Node ids in `nodes` are the
 * reference's 1-based Int64.  Parent indices returned are 1-based, 0 = baseline.
 */
#ifndef NHP_ORACLE_H
#define NHP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OK = 0, ORC_EINVAL = 1, ORC_EDOMAIN = 2, ORC_ENOMEM = 4 };
enum { ORC_BASELINE_HOMOGENEOUS = 0, ORC_BASELINE_LGCP = 1 };
enum { ORC_IMPULSE_EXPONENTIAL = 0, ORC_IMPULSE_LOGITNORMAL = 1 };
/* `flags` argument: bit0 selects the arithmetic (0 = libm, reference-faithful formulas;
 * 1 = the deterministic bit-contract sequences of nhp_detmath.h that the HIP kernels
 * restate); bit1 = hoist the per-event row sums of W out of the event loop (same
 * mathematics, used only to make the timed CPU baseline a stronger one). */
enum { ORC_MATH_LIBM = 0, ORC_MATH_DET = 1, ORC_FAST_INTEGRAL = 2 };

typedef struct {
    int32_t n_nodes;
    int32_t baseline_kind;
    const double *lambda0;  /* homogeneous: [N]; LGCP: [N*grid_n], node c at c*grid_n */
    const double *grid_x;   /* LGCP grid, [grid_n] */
    int32_t grid_n;
    int32_t impulse_kind;
    const double *theta;    /* exponential rate, [N*N] */
    const double *mu;       /* logit-normal location, [N*N] */
    const double *tau;      /* logit-normal precision, [N*N] */
    double dt_max;
    const double *W;        /* [N*N] */
    const double *A;        /* adjacency (0/1 as double) or NULL for the standard process */
} orc_cont_model;

/* ---- evaluators (src/impulses.jl:106-108,174-178; src/baselines.jl:98-118,328-336) */
double orc_impulse_exponential(double theta, double dt, int flags);
double orc_impulse_logitnormal(double mu, double tau, double dt_max, double dt, int flags);
int orc_baseline_intensity(const orc_cont_model *m, int64_t node1, double t, double *out);
int orc_baseline_integral(const orc_cont_model *m, double duration, double *out_per_node);
int orc_linear_interpolate(const double *x, const double *y, int32_t n, double x0, double *out);
double orc_linear_integrate(const double *x, const double *y, int32_t n);

/* ---- continuous log-likelihood (src/continuous.jl:210-305,360-442,521-531) */
int orc_cont_loglik_windowed(const orc_cont_model *m, const double *times, const int64_t *nodes,
                             int64_t M, double duration, int flags, double *ll);
/* threaded branch (src/continuous.jl:224-232); all-cores CPU baseline */
int orc_cont_loglik_windowed_mt(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                int64_t M, double duration, int flags, int threads, double *ll_out);
int orc_max_threads(void);
int orc_cont_loglik_recursive(const orc_cont_model *m, const double *times, const int64_t *nodes,
                              int64_t M, double duration, int flags, double *ll);
/* per-event total intensity, lambda[i] for i in [i0,i1) (src/continuous.jl:286-300) */
int orc_cont_total_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                             int64_t M, int64_t i0, int64_t i1, int flags, double *lambda);
/* intensity(process, data, times) -> Q x N column-major (src/continuous.jl:76-96) */
int orc_cont_intensity(const orc_cont_model *m, const double *times, const int64_t *nodes,
                       int64_t M, const double *q, int64_t Q, int flags, double *out);
int64_t orc_cont_pair_count(const double *times, int64_t M, double dt_max);

/* ---- parent sampler + Gibbs sufficient statistics (src/parents.jl:1-79; etc.) */
int orc_cont_resample_parents(const orc_cont_model *m, const double *times, const int64_t *nodes,
                              int64_t M, const double *u, int flags,
                              int64_t *parents, int64_t *parentnodes);
void orc_uniform_stream(uint64_t seed, uint64_t step, int64_t M, double *u);
void orc_node_counts(const int64_t *nodes, int64_t M, int32_t N, double *Mn);
void orc_parent_counts(const int64_t *nodes, const int64_t *parentnodes, int64_t M, int32_t N, double *Mnm);
void orc_baseline_node_counts(const int64_t *nodes, const int64_t *parentnodes, int64_t M, int32_t N, double *cnt0);
void orc_duration_mean(const double *times, const int64_t *nodes, const int64_t *parents,
                       int64_t M, int32_t N, double *Xnm);
void orc_log_duration_stats(const double *times, const int64_t *nodes, const int64_t *parents,
                            int64_t M, int32_t N, double dt_max, double *Xnm, double *Vnm);

/* ---- adjacency-matrix Gibbs sweep (src/continuous.jl:444-519; logsumexp src/utils/helpers.jl:13-16;
 * link_probability src/networks.jl:65-68).  A (N*N, 0/1 doubles) is updated in place, column by
 * column, parent by parent; u[p + c*N] is the explicit uniform of the Bernoulli draw
 * ([3P] Distributions: rand(Bernoulli(q)) = rand() <= q); rho[p + c*N] the link probability. */
int orc_cont_resample_adjacency(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                int64_t M, double duration, const double *rho, const double *u, double *A);
/* the same sweep restricted to columns [c0, c1) (columns are independent, src/continuous.jl:460-487) */
int orc_cont_resample_adjacency_columns(const orc_cont_model *m, const double *times, const int64_t *nodes,
                                        int64_t M, double duration, const double *rho, const double *u, double *A,
                                        int32_t c0, int32_t c1);

/* ---- discrete Gibbs parent counts (src/parents.jl:82-134), counts[c + N*k], k = 0 baseline, 1 + p*B + b */
int orc_disc_resample_parents(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                              const double *lambda0, const double *W, const double *theta, const double *A,
                              double dt, uint64_t seed, uint64_t step, int64_t *counts);

/* ---- DiscreteLogGaussianCoxProcess baseline (src/baselines.jl:461-609) */
void orc_disc_intensity_b(const double *conv, int64_t T, int32_t N, int32_t B, const double *lambda0,
                          const double *base_tn, const double *W, const double *theta, const double *A, double dt,
                          double *lam);
int orc_disc_lgcp_intensity(const double *x, int32_t G, const double *lam, int32_t N, double dt,
                            const double *times, int64_t ntimes, double *out);
int orc_disc_resample_parents_b(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                                const double *lambda0, const double *base_tn, const double *W, const double *theta,
                                const double *A, double dt, uint64_t seed, uint64_t step, int64_t *counts,
                                int64_t *base_counts);
int orc_disc_lgcp_loglik(const int64_t *s0, int64_t T, int32_t N, const double *x, int32_t G, const double *cand,
                         double dt, double *ll);

/* ---- discrete adjacency Gibbs sweep (src/discrete.jl:424-480), literal; A [N*N] in place */
int orc_disc_resample_adjacency(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                                const double *lambda0, const double *W, const double *theta, double dt,
                                const double *rho, const double *u, double *A);

/* ---- LGCP baseline likelihood inside the elliptical-slice sampler (src/baselines.jl:227-254) */
int orc_lgcp_loglik(const double *times, const int64_t *nodes, const int64_t *parentnodes, int64_t M,
                    int32_t N, const double *grid_x, int32_t G, const double *lam, double *ll);

/* ---- analytic gradient of the continuous ll (formulas: SURVEY.md 7; no reference code) */
int orc_cont_loglik_grad(const orc_cont_model *m, const double *times, const int64_t *nodes,
                         int64_t M, double duration, int recursive, double *ll, double *grad);

/* ---- discrete path (src/discrete.jl:86-151,369-385; src/impulses.jl:321-335; src/parents.jl:136-177) */
int orc_disc_basis(int32_t L, int32_t B, double dt, double *phi /* [L*B], lag fastest */);
void orc_disc_convolve(const int64_t *data, int32_t N, int64_t T, const double *phi, int32_t L,
                       int32_t B, double *conv /* [T*N*B], t fastest */);
void orc_disc_intensity(const double *conv, int64_t T, int32_t N, int32_t B, const double *lambda0,
                        const double *W, const double *theta, const double *A, double dt,
                        double *lam /* [T*N] */);
double orc_disc_loglik(const int64_t *data, const double *lam, int64_t T, int32_t N);
double orc_digamma(double x);
/* one VB update! step; all variational arrays are updated in place */
int orc_disc_vb_step(const int64_t *data, const double *conv, int64_t T, int32_t N, int32_t B,
                     double dt, double alpha0, double beta0, double kappa, double nu, double gamma,
                     double *alpha_v, double *beta_v, double *kappa_v, double *nu_v, double *gamma_v);

/* ---- det-math probes for the bitwise contract tests */
double orc_det_exp(double x);
double orc_det_log(double x);

#ifdef __cplusplus
}
#endif
#endif
