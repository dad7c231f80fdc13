/*
 * include/nhp.h -- C ABI of libnhp.so, the MI355X (gfx950) hot path for network Hawkes
 * processes.
 *
 * The reference (cswaney/NetworkHawkesProcesses.jl v0.1.0) is pure Julia and has no FFI:
 * its "plug-in API" is multiple dispatch on Baseline / ImpulseResponse / Weights / Network
 * components composed into process structs (src/continuous.jl:108-112,315-321;
 * src/discrete.jl:161-170,395-402).  This header is the boundary a Julia shim binds with
 * `ccall`: each component lowers to plain arrays + a kind enum in
 * nhp_cont_model_desc, and each entry point below replaces the Julia method cited next to
 * it.  Citations are file:line in the reference checkout.
 *
 * Conventions
 *  - Julia layout is kept so the shim passes its arrays untouched: matrices are
 *    column-major, X[p,c] at p + c*N (p = parent node, c = child node,
 *    src/continuous.jl:303); theta[p,c,b] at p + c*N + b*N*N; data[n,t] at n + t*N;
 *    convolved[t,n,b] at t + n*T + b*T*N.
 *  - `nodes` are the reference's 1-based Int64 (src/continuous.jl:14).  Parent indices
 *    returned are 1-based event indices, 0 = baseline (src/parents.jl:41-45).
 *  - Host pointers are borrowed for the duration of the call only.  Device memory belongs
 *    to the ctx / dataset / model handles and is released by the *_destroy functions.
 *  - No exception crosses the boundary: every function returns an nhp_status; the message
 *    is available from nhp_last_error().  The shim maps NHP_EDOMAIN -> DomainError
 *    (src/baselines.jl:100,106,111,116), NHP_ESHAPE/NHP_EINVAL -> ErrorException
 *    (src/impulses.jl:44-45, src/weights.jl:10-11).
 *  - A ctx owns one HIP device and one stream and is not thread-safe; distinct ctx are
 *    independent (one host thread or process per GPU).
 */
#ifndef NHP_H
#define NHP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NHP_OK = 0,
    NHP_EINVAL = 1,   /* bad argument / unsorted events */
    NHP_EDOMAIN = 2,  /* negative time or duration, node id out of 1..N, non-positive intensity */
    NHP_ESHAPE = 3,   /* array length does not match the model */
    NHP_ENOMEM = 4,
    NHP_EHIP = 5,     /* HIP runtime error (no device, launch failure, ...) */
    NHP_ENOTIMPL = 6,
    NHP_ERCCL = 7     /* RCCL unavailable (librccl.so.1 not loadable) or a collective failed */
} nhp_status;

enum { NHP_BASELINE_HOMOGENEOUS = 0, NHP_BASELINE_LGCP = 1 };
enum { NHP_IMPULSE_EXPONENTIAL = 0, NHP_IMPULSE_LOGITNORMAL = 1 };

/* flags for nhp_cont_loglik* */
enum {
    NHP_LL_RECURSIVE = 1,        /* loglikelihood(...; recursive=true) for exponential impulses
                                    (src/continuous.jl:212-214,361-363): O(M*N) recursion that
                                    ignores dt_max; for other impulses the flag is ignored, as
                                    in the reference */
    NHP_LL_FULL_RECURSION = 2,   /* with NHP_LL_RECURSIVE: always run the O(M*N) recursion, never its
                                    truncated-window evaluation (same value to fp64 resolution) */
};

typedef struct nhp_ctx nhp_ctx;
typedef struct nhp_cont_dataset nhp_cont_dataset;
typedef struct nhp_cont_model nhp_cont_model;
typedef struct nhp_disc_dataset nhp_disc_dataset;
typedef struct nhp_comm nhp_comm;      /* RCCL communicator bound to a ctx (multi-GPU section at the end) */

/* The lowered form of (Baseline, ImpulseResponse, Weights, adjacency_matrix):
 *   HomogeneousProcess.λ                      src/baselines.jl:27-39
 *   LogGaussianCoxProcess.x / .λ              src/baselines.jl:148-173 (evaluator only)
 *   ExponentialImpulseResponse.θ / .Δtmax     src/impulses.jl:30-37
 *   LogitNormalImpulseResponse.μ / .τ / .Δtmax src/impulses.jl:138-148
 *   DenseWeightModel.W                        src/weights.jl:47-55
 *   ContinuousNetworkHawkesProcess.adjacency_matrix  src/continuous.jl:315-321 */
typedef struct {
    int32_t n_nodes;
    int32_t baseline_kind;
    const double *lambda0;   /* homogeneous: [N]; LGCP: [N*grid_n], node c at c*grid_n */
    const double *grid_x;    /* LGCP grid points [grid_n], strictly increasing, x[0] = 0 */
    int32_t grid_n;          /* 0 for the homogeneous baseline */
    int32_t impulse_kind;
    const double *theta;     /* exponential: [N*N] */
    const double *mu;        /* logit-normal: [N*N] */
    const double *tau;       /* logit-normal: [N*N] */
    double dt_max;           /* must equal the dataset's dt_max */
    const double *W;         /* [N*N] */
    const double *A;         /* [N*N] of 0.0/1.0, or NULL for the standard (dense) process */
} nhp_cont_model_desc;

/* Gibbs sufficient statistics emitted by the parent sampler in the same pass
 * (src/baselines.jl:87-96; src/parents.jl:61-79; src/impulses.jl:84-96,216-252).
 * Any pointer may be NULL.  All are [N] or [N*N] column-major doubles, as the reference
 * stores counts in Float64 `zeros`. */
typedef struct {
    double *cnt0;   /* [N]   baseline-attributed events per node */
    double *Mn;     /* [N]   events per node */
    double *Mnm;    /* [N*N] events on c attributed to a parent on p */
    double *Xnm;    /* [N*N] exponential: mean Δt (NaN -> 0); logit-normal: mean log(Δt/(Δtmax-Δt)) (NaN kept) */
    double *Vnm;    /* [N*N] logit-normal: Σ (log-duration - Xnm)^2; untouched for exponential */
} nhp_cont_stats;

/* ---- context ------------------------------------------------------------------------- */
nhp_status nhp_ctx_create(int32_t device, nhp_ctx **out);
void nhp_ctx_destroy(nhp_ctx *ctx);
const char *nhp_last_error(const nhp_ctx *ctx);      /* ctx may be NULL: last global error */
nhp_status nhp_ctx_synchronize(nhp_ctx *ctx);
/* hipEvent pair on the ctx stream, for measuring kernel time without host overhead */
nhp_status nhp_ctx_timer_start(nhp_ctx *ctx);
nhp_status nhp_ctx_timer_stop(nhp_ctx *ctx, double *elapsed_ms);
int32_t nhp_abi_version(void);
/* sizeof / offsetof of every struct that crosses this boundary, so a binding (ctypes Structure, Julia `struct`) can
 * assert its layout against the library it loaded instead of trusting a header it cannot include.  Writes up to `cap`
 * int32 entries into `out` and returns the number of entries the library has (NHP_ABI_LAYOUT_LEN):
 *   [0] sizeof(nhp_cont_model_desc), then offsetof n_nodes, baseline_kind, lambda0, grid_x, grid_n, impulse_kind, theta, mu,
 *       tau, dt_max, W, A                                                            (entries 1..12)
 *   [13] sizeof(nhp_gibbs_priors), then offsetof alpha0, beta0, kappa, nu, a, b, mu_mu, kappa_mu   (14..21)
 *   [22] sizeof(nhp_cont_stats), then offsetof cnt0, Mn, Mnm, Xnm, Vnm                             (23..27)
 *   [28] NHP_MAX_SLOTS, [29] NHP_COMM_ID_BYTES */
#define NHP_ABI_LAYOUT_LEN 30
int32_t nhp_abi_layout(int32_t *out, int32_t cap);

/* ---- continuous data: (events, nodes, duration)  src/continuous.jl:14,29-36 ----------- */
/* Validates (sorted, >= 0, nodes in 1..N, duration >= 0), runs the look-back pre-pass for
 * dt_max (window starts, node buckets, work partition) and uploads once. */
nhp_status nhp_cont_dataset_create(nhp_ctx *ctx, const double *events, const int64_t *nodes,
                                   int64_t n_events, int32_t n_nodes, double duration,
                                   double dt_max, nhp_cont_dataset **out);
/* Column shard of the same data for evaluating ONE log-likelihood / gradient on several GPUs (second
 * way): the log-likelihood (src/continuous.jl:216-237) is a sum over child nodes c of
 *   -∫λ0_c - Σ_p cnt[p]·[A·]W[p,c] + Σ_{i: c_i = c} log λ_i
 * and the gradient is block-separable in the same columns.  The shard holds every event (all of them are parents) but
 * evaluates only the children on the 0-based nodes [col_begin, col_end); nhp_cont_loglik / _enqueue / _batch / _grad
 * on it return that part (gradient entries of other columns are 0), so the parts of a partition of [0, n_nodes) add up
 * to the whole -- one scalar (or P-vector) all-reduce per evaluation.  The Gibbs sweep is separable in the same way
 * (parents of the children on c, the statistics and conjugate draws of column c and the sweep of A[:, c] involve column c
 * only, and every random stream is keyed by global event / entry indices), so nhp_cont_resample_parents,
 * nhp_cont_gibbs_step and nhp_cont_resample_adjacency on a shard update exactly their columns -- with the same values
 * a whole-dataset sweep gives them -- and leave the rest untouched (returned parents / statistics of other columns are
 * 0; n_links counts the shard's links).  nhp_cont_event_intensity and nhp_cont_lgcp_loglik need every column and
 * return NHP_ENOTIMPL on a shard. */
nhp_status nhp_cont_dataset_create_columns(nhp_ctx *ctx, const double *events, const int64_t *nodes,
                                           int64_t n_events, int32_t n_nodes, double duration, double dt_max,
                                           int32_t col_begin, int32_t col_end, nhp_cont_dataset **out);
void nhp_cont_dataset_destroy(nhp_cont_dataset *ds);
/* Σ_i K_i: parent-child pairs inside the look-back window (SURVEY 8d F_alg) */
int64_t nhp_cont_dataset_pairs(const nhp_cont_dataset *ds);

/* ---- continuous model: device-resident parameter blob --------------------------------- */
nhp_status nhp_cont_model_create(nhp_ctx *ctx, const nhp_cont_model_desc *desc, nhp_cont_model **out);
/* re-upload all parameters (same kinds / shapes as at creation) */
nhp_status nhp_cont_model_update(nhp_ctx *ctx, nhp_cont_model *model, const nhp_cont_model_desc *desc);
/* params!(process, x) for the standard process: x = [λ0; θ | μ; τ; W]  src/continuous.jl:121-129 */
nhp_status nhp_cont_model_set_params(nhp_ctx *ctx, nhp_cont_model *model, const double *x, int64_t len);
void nhp_cont_model_destroy(nhp_cont_model *model);

/* ---- loglikelihood(process, data; recursive)  src/continuous.jl:210-276,360-442 -------- */
nhp_status nhp_cont_loglik(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *model,
                           int32_t flags, double *ll);
/* Asynchronous form for batches (finite-difference sweeps, chains): enqueue evaluations
 * into result slots [0, NHP_MAX_SLOTS), then fetch them with one synchronisation. */
#define NHP_MAX_SLOTS 4096
nhp_status nhp_cont_loglik_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                   const nhp_cont_model *model, int32_t flags, int32_t slot);
nhp_status nhp_ctx_fetch(nhp_ctx *ctx, int32_t first_slot, int32_t n, double *out);
/* nb log-likelihoods of nb device-resident models on one dataset (the 2P objective calls of a
 * finite-difference gradient, a population of chains), one synchronisation.  The evaluations are independent, so at
 * short windows compatible models share one pass over the data (up to four per launch) and the launches alternate
 * between the context's two internal streams; every result is the same as from nhp_cont_loglik on that model.
 * NHP_BATCH_FUSE=1 / NHP_BATCH_LANES=1 in the environment turn either off. */
nhp_status nhp_cont_loglik_batch(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                 const nhp_cont_model *const *models, int32_t nb, int32_t flags, double *ll);

/* log-likelihood and its analytic gradient in params! order [λ0; θ | μ; τ; W] (homogeneous
 * baseline).  Replaces the 2P finite-difference objective calls Optim makes inside mle!
 * (src/continuous.jl:144-198). */
nhp_status nhp_cont_loglik_grad(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *model,
                                int32_t flags, double *ll, double *grad, int64_t grad_len);

/* total_intensity at every event, λ_{c_i}(t_i)  src/continuous.jl:286-300,391-405 */
nhp_status nhp_cont_event_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                    const nhp_cont_model *model, double *lambda /* [M] */);

/* intensity(process, data, times) -> Q x N column-major  src/continuous.jl:76-96 */
nhp_status nhp_cont_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *model,
                              const double *times, int64_t n_times, double *out);

/* resample_parents(process, data)  src/parents.jl:1-46.  One categorical draw per event from
 * an explicit uniform stream: `u` (host, [M]) if non-NULL, else Philox4x32-10 keyed by
 * (seed, step, event index).  parents / parentnodes may be NULL (statistics only). */
nhp_status nhp_cont_resample_parents(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                     const nhp_cont_model *model, const double *u,
                                     uint64_t seed, uint64_t step,
                                     int64_t *parents, int64_t *parentnodes, nhp_cont_stats *stats);
/* One Gibbs sweep of resample!(process, data) (src/continuous.jl:202-208,350-358) entirely on the
 * device: parents + statistics as above, then the conjugate draws of HomogeneousProcess
 * (src/baselines.jl:72-77), DenseWeightModel (src/weights.jl:59-64) and the impulse response
 * (src/impulses.jl:68-73 or :204-214) written straight into the device-resident model.  Draws are
 * Philox-keyed by (seed, step, element): reproducible; distributionally (not bitwise) equal to
 * Julia's samplers.  The adjacency matrix is left unchanged. */
typedef struct {
    double alpha0, beta0;    /* HomogeneousProcess.α0, β0 */
    double kappa, nu;        /* DenseWeightModel.κ, ν */
    double a, b;             /* ExponentialImpulseResponse.α, β  |  LogitNormalImpulseResponse.α0, β0 */
    double mu_mu, kappa_mu;  /* LogitNormalImpulseResponse.μμ, κμ */
} nhp_gibbs_priors;
/* Asynchronous: the sweep is enqueued and the call returns without draining the GPU, so a chain keeps one sweep in
 * flight.  The sampler's only run-time failure (the weights of some event do not sum to a positive finite value, the
 * error nhp_cont_resample_parents reports at once) therefore surfaces as NHP_EDOMAIN from the NEXT nhp_cont_gibbs_step
 * on this context, or from the next call that synchronises it (nhp_ctx_synchronize, the parameter / moment / result
 * downloads), once. */
nhp_status nhp_cont_gibbs_step(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *model,
                               const nhp_gibbs_priors *priors, uint64_t seed, uint64_t step);
/* resample_adjacency_matrix!(process, data)  src/continuous.jl:444-487: one Gibbs sweep over the
 * N x N adjacency matrix of the device-resident model (updated in place).  Link probabilities
 * (src/networks.jl:65-68): `rho_matrix` [N*N] if non-NULL, else the scalar `rho`.  Bernoulli draws
 * (u <= p, as Distributions.jl) use `u` [N*N] if non-NULL, else Philox keyed (seed, step, p + c*N).
 * A_out (nullable) receives the new matrix, n_links (nullable) its number of ones (the statistic
 * BernoulliNetworkModel's resample! needs, src/networks.jl:70-78). */
nhp_status nhp_cont_resample_adjacency(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *model,
                                       const double *rho_matrix, double rho, const double *u,
                                       uint64_t seed, uint64_t step, double *A_out, double *n_links);
/* loglikelihood(process::LogGaussianCoxProcess, data, node, y)  src/baselines.jl:247-254, for all
 * nodes in one call: ll[c] = -trapezoid(lam_c) + Σ log lam_c(t_i) over node c's events that
 * split_extract (:227-238) attributes to the baseline (sampled parent node 0).  lam [N*grid_n]
 * (node c at c*grid_n) is the candidate intensity exp.(m .+ y_c) on grid_x [grid_n].  The
 * attribution is `parentnodes` [M] (resample_parents' second vector) if non-NULL -- it then
 * stays on the device with the dataset -- else the one left there by the latest
 * nhp_cont_resample_parents / nhp_cont_gibbs_step on this dataset.  NHP_EDOMAIN if an event lies
 * outside [grid_x[0], grid_x[end]] (the interpolator's DomainError, src/utils/interpolation.jl:27). */
nhp_status nhp_cont_lgcp_loglik(nhp_ctx *ctx, const nhp_cont_dataset *ds, const int64_t *parentnodes,
                                const double *grid_x, int32_t grid_n, const double *lam, double *ll);
/* params(process) of the device-resident model: [λ0; θ | μ; τ; W]  src/continuous.jl:116-119 */
nhp_status nhp_cont_model_get_params(nhp_ctx *ctx, const nhp_cont_model *model, double *x, int64_t len);
/* process.adjacency_matrix of the device-resident model (after nhp_cont_network_step / nhp_cont_mcmc_run): [N*N] 0.0/1.0 */
nhp_status nhp_cont_model_get_adjacency(nhp_ctx *ctx, const nhp_cont_model *model, double *A, int64_t len);
/* Sample store on the device.  mcmc! appends params(process) after every sweep (src/inference.jl:61):
 * 4N²+N doubles per step, 33.5 MB at N = 1024 -- more PCIe time than the sweep itself.  These keep Σx and Σx² of the
 * device-resident parameters instead: _reset zeroes them, _accumulate adds the current [λ0; θ | μ; τ; W; vec(A) if the
 * model has an adjacency matrix] (call it once per sweep), _fetch returns the two sums and the number of samples;
 * len = N + N²·(1 | 2) + N² [+ N²].  Posterior mean = sum / count, second moment = sumsq / count. */
nhp_status nhp_cont_model_moments_reset(nhp_ctx *ctx, nhp_cont_model *model);
nhp_status nhp_cont_model_moments_accumulate(nhp_ctx *ctx, nhp_cont_model *model);
nhp_status nhp_cont_model_moments_fetch(nhp_ctx *ctx, const nhp_cont_model *model, double *sum, double *sumsq,
                                        int64_t len, int64_t *count);
/* the uniform stream itself (host side, same bits as the kernel draws) */
void nhp_uniform_stream(uint64_t seed, uint64_t step, int64_t n, double *u);

/* diagnostics: evaluate a device math primitive elementwise (op 0 exp, 1 log, 2 sqrt, 3 x/y,
 * 4 exp for x<=0, 5 exponential pdf(θ=x, Δt=y), 6 logit-normal pdf(τ=x, Δt=y; μ=.25, Δtmax=2), 7 the log-likelihood
 * kernels' table-driven exp for x<=0 (nhp_exp_neg_tab: within 2 ulp, not part of the bitwise contract), 8 θ·e^{-θΔt} through it);
 * lets tests hold the kernels' fixed operation sequences to a bitwise contract */
nhp_status nhp_probe_math(nhp_ctx *ctx, int32_t op, const double *x, const double *y, int64_t n, double *out);
/* diagnostics: n device-side random variates with the generators and Philox keying of the Gibbs kernels (nhp_rng.h):
 * kind 0 Gamma(shape a[i], scale b[i]) keyed (seed, step, i); 1 standard normal keyed (seed, step, i); 2 Beta(a[i], b[i]) as
 * X/(X+Y) from Gammas keyed (seed, step, 2i) and (seed, step, 2i+1).  Lets tests hold the draws to their distributions
 * (Kolmogorov-Smirnov) and to known answers of the Philox -> uniform -> variate chain. */
nhp_status nhp_probe_draws(nhp_ctx *ctx, int32_t kind, uint64_t seed, uint64_t step, int64_t n, const double *a, const double *b,
                           double *out);
/* throughput calibration on register operands: mode 0 = exponential pair terms per second (the
 * fp64-VALU ceiling of the windowed kernels), mode 1 = fp64 fma per second */
nhp_status nhp_probe_rate(nhp_ctx *ctx, int32_t mode, int32_t iters, int32_t blocks, double *ops_per_s);
/* gather calibration: n_windows scattered windows of `recs` 16-byte records out of an array of array_recs records,
 * 8 lanes per window, no arithmetic -- microseconds per launch (the floor of the short-window kernels) */
nhp_status nhp_probe_gather(nhp_ctx *ctx, int32_t n_windows, int32_t recs, int64_t array_recs, int32_t blocks,
                            double *us_per_launch);
/* streaming-read calibration: `blocks` workgroups of `threads` (256 | 512) sweep contiguous shares of a `bytes`-byte buffer,
 * mode 0 = 16 bytes per lane, mode 1 = the child slices' two planes (4 + 2 bytes per lane) -- microseconds per launch
 * (launched back to back: below 256 MB the Infinity Cache serves it, as it does the repeated log-likelihood) */
nhp_status nhp_probe_stream(nhp_ctx *ctx, int32_t mode, int64_t bytes, int32_t blocks, int32_t threads, double *us_per_launch,
                            int64_t *bytes_read /* nullable: what a launch really reads (whole rows per wave) */);
/* the mle! optimizer alone (csrc/nhp_lbfgs.h: the projected L-BFGS both nhp_cont_mle_run and nhp_disc_mle_run drive) on the
 * separable quadratic f(x) = 1/2 sum_i h[i] (x[i] - c[i])^2 over the box [lower, upper]^n, host vectors h, c and x (start in,
 * minimiser out): step counts that can be held against another L-BFGS without a likelihood in between */
nhp_status nhp_probe_lbfgs(nhp_ctx *ctx, int64_t n, const double *h, const double *c, double lower, double upper, double f_abstol,
                           int32_t max_steps, double *x, double *loss, int32_t *steps, int32_t *converged, int32_t *evaluations);

/* ---- discrete data: N x T counts  src/discrete.jl:18,80 -------------------------------- */
nhp_status nhp_disc_dataset_create(nhp_ctx *ctx, const int64_t *data, int32_t n_nodes, int64_t n_bins,
                                   nhp_disc_dataset **out);
void nhp_disc_dataset_destroy(nhp_disc_dataset *ds);
/* basis(impulse)  src/impulses.jl:321-335 -> phi [L*B], lag fastest (host) */
nhp_status nhp_disc_basis(int32_t n_lags, int32_t n_basis, double dt, double *phi);
/* convolve(process, data)  src/discrete.jl:146-151; keeps the T x N x B result on the device
 * and copies it to `out` if non-NULL */
nhp_status nhp_disc_convolve(nhp_ctx *ctx, nhp_disc_dataset *ds, const double *phi, int32_t n_lags,
                             int32_t n_basis, double *out);
/* intensity(process, convolved)  src/discrete.jl:115-129 -> T x N; A may be NULL */
nhp_status nhp_disc_intensity(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                              const double *W, const double *theta, const double *A, double dt,
                              double *lam);
/* loglikelihood(process, data, convolved)  src/discrete.jl:91-102 */
nhp_status nhp_disc_loglik(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                           const double *W, const double *theta, const double *A, double dt,
                           double *ll);
/* loglikelihood(process, data, convolved) and its gradient in mle!'s parameter vector
 * [λ0 (N); vec(W .* θ) (N*N*B)]  (params / params! src/discrete.jl:174-201): what the 2P finite-difference
 * objective calls per gradient inside mle! (src/discrete.jl:211-296) are replaced by */
nhp_status nhp_disc_loglik_grad(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                const double *W, const double *theta, double dt, double *ll, double *grad,
                                int64_t grad_len);
/* one update!(process, data, convolved) mean-field step  src/discrete.jl:369-375;
 * variational parameters are read and overwritten in place (host arrays) */
nhp_status nhp_disc_vb_step(nhp_ctx *ctx, const nhp_disc_dataset *ds, double dt,
                            double alpha0, double beta0, double kappa, double nu, double gamma,
                            double *alpha_v, double *beta_v, double *kappa_v, double *nu_v,
                            double *gamma_v);
/* Parent counts of one discrete Gibbs sweep: resample_parents(process, data, convolved)
 * src/parents.jl:82-116 summed over time, counts[c + N*k] = Σ_t parents[t, c, k] with k = 0 the
 * baseline and k = 1 + p*B + b (0-based p, b) parent node p through basis b -- the statistic the
 * discrete resample! methods consume (src/baselines.jl:413-419, src/weights.jl:28-35,
 * src/impulses.jl:337-353).  A bin's Multinomial draw is taken as n categorical draws through
 * explicit Philox uniforms keyed (seed, step, bin, event): same distribution as Distributions.jl's
 * sampler, reproducible, and equal to the oracle bit for bit.  counts: [N * (1 + N*B)] int64. */
nhp_status nhp_disc_resample_parents(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                     const double *W, const double *theta, const double *A, double dt,
                                     uint64_t seed, uint64_t step, int64_t *counts);
/* resample!(process::DiscreteStandardHawkesProcess, data, convolved)  src/discrete.jl:362-368 in one call: the
 * parent counts above, then the conjugate draws on the device (Philox-keyed; distributional parity with Julia):
 * λ0 ~ Gamma(α0 + counts[:, 0], 1/(β0 + T dt)) (the intended form of src/baselines.jl:413-419),
 * W ~ Gamma(κ + Σ_b counts, 1/(ν + Σ_t data[p, :]))  src/weights.jl:59-64,
 * θ[p, c, :] ~ Dirichlet(γ + counts)  src/impulses.jl:337-353.  lambda0, W, theta are read and overwritten. */
nhp_status nhp_disc_gibbs_step(nhp_ctx *ctx, const nhp_disc_dataset *ds, double *lambda0, double *W, double *theta,
                               const double *A, double dt, double alpha0, double beta0, double kappa, double nu,
                               double gamma0, uint64_t seed, uint64_t step);
/* resample_adjacency_matrix!(process::DiscreteNetworkHawkesProcess, data, convolved)
 * src/discrete.jl:424-480: one Gibbs sweep over A [N*N] (host, updated in place), columns in parallel,
 * entries of a column in sequence, each conditional on the current column.  Link probabilities
 * (src/networks.jl:65-68): rho_matrix [N*N] if non-NULL, else the scalar rho; Bernoulli draws (u <= q, as
 * Distributions.jl) from u [N*N] if non-NULL, else Philox keyed (seed, step, p + c*N).  n_links
 * (nullable) receives ΣA for BernoulliNetworkModel's resample! (src/networks.jl:70-78). */
nhp_status nhp_disc_resample_adjacency(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                       const double *W, const double *theta, double *A, double dt,
                                       const double *rho_matrix, double rho, const double *u,
                                       uint64_t seed, uint64_t step, double *n_links);
/* DiscreteLogGaussianCoxProcess(x, λ, Σ, m, dt) as this dataset's baseline (src/baselines.jl:461-509): lam
 * [grid_n * N] (λ[:, n] at n*grid_n) on grid_x [grid_n].  The per-bin baseline intensity(p, 1:T)
 * (src/discrete.jl:117, src/baselines.jl:531-537) is built on the device and kept with the dataset; the other
 * nhp_disc_* calls use it when their lambda0 argument is NULL (nhp_disc_loglik_grad then returns the gradient
 * in [vec(λ) (grid_n*N); vec(W .* θ)] order).  NHP_EDOMAIN if a bin time 1..T lies outside the grid. */
nhp_status nhp_disc_set_lgcp_baseline(nhp_ctx *ctx, nhp_disc_dataset *ds, const double *grid_x, int32_t grid_n,
                                      const double *lam, double dt);
/* loglikelihood(p::DiscreteLogGaussianCoxProcess, data, node, y)  src/baselines.jl:571-584 for all nodes:
 * data = parents[:, :, 1] of the latest nhp_disc_resample_parents on this dataset (kept on the device),
 * cand [grid_n * N] = exp.(m .+ y) per node -- the body of every elliptical_slice round (:640-679). */
nhp_status nhp_disc_lgcp_loglik(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *cand, double dt, double *ll);
/* n_steps consecutive update! steps of vb! (src/inference.jl:153-181) with the variational parameters
 * resident on the device in between (one upload, one download) */
nhp_status nhp_disc_vb_run(nhp_ctx *ctx, const nhp_disc_dataset *ds, double dt,
                           double alpha0, double beta0, double kappa, double nu, double gamma, int32_t n_steps,
                           double *alpha_v, double *beta_v, double *kappa_v, double *nu_v, double *gamma_v);

/* ---- several GPUs: RCCL over xGMI   ------------------------------------------------------------
 * One process (or host thread) per GPU, one nhp_ctx each.  The reference has no distributed code (README.md:42 lists
 * "multiple-trial inference" as future work; mcmc! has no cross-chain term, src/inference.jl:49-70), so these entry
 * points replace nothing: they are the exchange steps of the two ways the path shards --
 *   independent units (chains, restarts): no data-path collective, one gather of per-chain summaries at the end;
 *   one evaluation / one chain over all ranks by child-node column: one all-reduce per evaluation / per step --
 * and they hand RCCL DEVICE pointers: results are reduced where the kernels left them, on the ctx stream, and cross
 * PCIe once, reduced.  librccl.so.1 is opened on first use (dlopen): a single-GPU host needs no RCCL installed.
 * Rendezvous: rank 0 calls nhp_comm_unique_id and gives the 128 bytes to the other ranks through whatever channel
 * the host has (Julia: Distributed / a file / MPI.bcast; Python: torch.distributed or a TCP store); every rank then
 * calls nhp_comm_create -- collectively, it blocks until all `world` ranks have joined. */
#define NHP_COMM_ID_BYTES 128
nhp_status nhp_comm_unique_id(uint8_t *id /* [NHP_COMM_ID_BYTES] */);
nhp_status nhp_comm_create(nhp_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, nhp_comm **out);
void nhp_comm_destroy(nhp_comm *comm);
int32_t nhp_comm_rank(const nhp_comm *comm);
int32_t nhp_comm_world(const nhp_comm *comm);
/* host vectors through a device staging buffer (small control data: link counts, log-likelihood traces, flags) */
nhp_status nhp_allreduce_sum(nhp_ctx *ctx, nhp_comm *comm, double *x, int64_t n);                 /* in place */
nhp_status nhp_allgather(nhp_ctx *ctx, nhp_comm *comm, const double *mine, int64_t n, double *all /* [world*n] */);
/* loglikelihood(process, data; recursive) by all ranks together: `ds` is this rank's column shard
 * (nhp_cont_dataset_create_columns); the partial result is all-reduced in place on the device and fetched once.
 * Every rank returns the same value. */
nhp_status nhp_cont_loglik_allreduce(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, const nhp_cont_model *model,
                                     int32_t flags, double *ll);
/* the mle! objective and its gradient (nhp_cont_loglik_grad) over all ranks: each rank's gradient is exact zeros outside
 * its columns, so the sum is the gradient; [ll; grad] is all-reduced on the device (P+1 doubles, 16.8 MB at N = 1024)
 * before the one download. */
nhp_status nhp_cont_loglik_grad_allreduce(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, const nhp_cont_model *model,
                                          int32_t flags, double *ll, double *grad, int64_t grad_len);
/* BASELINE config 5: the per-chain posterior summaries (the running sums of nhp_cont_model_moments_*, still on each
 * rank's device) all-gathered over RCCL: sum_all / sumsq_all [world * len] (rank r at r*len), counts [world],
 * rho_all [world * 3] (ρ, Σρ, Σρ² of nhp_cont_model_get_rho; zeros for a model without a device-side ρ). */
nhp_status nhp_gather_moments(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_model *model, double *sum_all, double *sumsq_all,
                              int64_t len, int64_t *counts, double *rho_all);

/* ---- network model on the device + chain driver  (src/networks.jl:54-78, src/inference.jl:49-70) ---------------
 * BernoulliNetworkModel.ρ kept next to the model on the device so that a network mcmc! step never drains the stream:
 * _set_rho uploads it, _get_rho returns {ρ, Σρ, Σρ²} (the sums follow nhp_cont_model_moments_accumulate / _reset). */
nhp_status nhp_cont_model_set_rho(nhp_ctx *ctx, nhp_cont_model *model, double rho);
nhp_status nhp_cont_model_get_rho(nhp_ctx *ctx, const nhp_cont_model *model, double *out /* [3] */);
/* resample_adjacency_matrix!(process, data) with the device-resident ρ (src/continuous.jl:444-487), then
 * resample!(network, A): ρ ~ Beta(α + ΣA, β + N² - ΣA) (src/networks.jl:70-78) drawn on the device as X/(X+Y) from two
 * Philox-keyed Gammas.  Asynchronous.  With a communicator, `ds` is a column shard: the shards' link counts are
 * all-reduced on the device and every rank draws the same ρ (same Philox key). */
/* alpha = beta = 0: ρ is held fixed (DenseNetworkModel, src/networks.jl:13-31: set ρ = 1). */
nhp_status nhp_cont_network_step(nhp_ctx *ctx, nhp_comm *comm /* nullable */, const nhp_cont_dataset *ds, nhp_cont_model *model,
                                 double alpha, double beta, uint64_t seed, uint64_t step);
/* The same step in two halves, for a host that exchanges the shards' link counts itself (no RCCL clique: ranks sharing
 * one GPU, a CPU-side rehearsal): _sweep runs the adjacency sweep with the device-resident ρ and returns this dataset's
 * link count (synchronises); _rho draws ρ ~ Beta(alpha + n_links, beta + n_entries - n_links) on the device with the same
 * Philox key as nhp_cont_network_step, so both routes give the same chain. */
nhp_status nhp_cont_network_sweep(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *model, uint64_t seed, uint64_t step,
                                  double *n_links);
nhp_status nhp_cont_network_rho(nhp_ctx *ctx, nhp_cont_model *model, double alpha, double beta, double n_links, double n_entries,
                                uint64_t seed, uint64_t step);
/* The body of mcmc!(process, data; nsteps) (src/inference.jl:55-62) for steps [step0, step0 + n_steps): per step one
 * nhp_cont_gibbs_step, for a network model one nhp_cont_network_step (net_alpha, net_beta = the Beta prior of ρ), and --
 * from chain step `burn` on -- one nhp_cont_model_moments_accumulate in place of push!(res.samples, params(process)).
 * Nothing crosses PCIe and the host synchronises once, at the end (where a sampler error of any step is reported).
 * comm (nullable): ONE chain swept by all ranks, each its column shard. */
nhp_status nhp_cont_mcmc_run(nhp_ctx *ctx, nhp_comm *comm /* nullable */, const nhp_cont_dataset *ds, nhp_cont_model *model,
                             const nhp_gibbs_priors *priors, double net_alpha, double net_beta, uint64_t seed,
                             uint64_t step0, int64_t n_steps, int64_t burn);

/* mle!(process, data; f_abstol, guess) (src/continuous.jl:144-198) with the optimizer's state on the device: minimises
 * -loglikelihood(process, data) over params(process) = [λ0 | grid intensities; θ | μ, τ; W] on the reference's box
 * [lower, upper]^P (1e-6, 10: src/continuous.jl:185-186) by projected L-BFGS fed the analytic gradient; the iterate,
 * gradient and history stay in HBM, the host reads scalars.  Stops by the reference's callback rule
 * |f_k - f_{k-1}| < f_abstol (src/continuous.jl:168-181) or at a stationary point of the box problem (*converged = 1),
 * after max_steps iterations or when no step decreases the objective (*converged = 0).  x [P]: the guess on entry
 * (clamped to the box), the minimiser on return; the device-resident `model` holds it too (params!(process, x)).
 * flags: NHP_LL_RECURSIVE as for nhp_cont_loglik (the reference's objective calls loglikelihood with its default).
 * comm (nullable): `ds` is this rank's column shard, [ll; ∇ll] is summed over the ranks on the device per evaluation and
 * every rank runs the same iteration.  *evals (nullable): objective + gradient evaluations spent.  Standard process only. */
nhp_status nhp_cont_mle_run(nhp_ctx *ctx, nhp_comm *comm /* nullable */, const nhp_cont_dataset *ds, nhp_cont_model *model, int32_t flags,
                            double lower, double upper, double f_abstol, int32_t max_steps, double *x, int64_t len,
                            double *loss, int32_t *steps, int32_t *converged, int32_t *evals);

/* mle!(process::DiscreteStandardHawkesProcess, data; f_abstol, guess) (src/discrete.jl:211-296) the same way: x = params(process)
 * = [λ0; vec(W .* θ)] (src/discrete.jl:178-182; homogeneous baseline), the objective -loglikelihood(process, data, convolved) with
 * params!'s split W = Σ_b, θ = x ./ W (:195-203) redone on the device per evaluation, its gradient from the two GEMMs of
 * nhp_disc_loglik_grad, the box and the |f_k - f_{k-1}| < f_abstol rule of the callback (:247-258; a monotone line search never
 * meets its loss-increase rule).  nhp_disc_convolve must have run on `data`. */
nhp_status nhp_disc_mle_run(nhp_ctx *ctx, const nhp_disc_dataset *data, double dt, double lower, double upper, double f_abstol,
                            int32_t max_steps, double *x, int64_t len, double *loss, int32_t *steps, int32_t *converged, int32_t *evals);

#ifdef __cplusplus
}
#endif
#endif
