// Internal declarations shared by the translation units of libnhp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/nhp.h"

#define NHP_BLOCK 256          // 4 waves of 64 lanes
#define NHP_WAVES (NHP_BLOCK / 64)
// children a group keeps in flight in the windowed kernels (host sorting and kernel must agree)
#ifndef NHP_U_SMALL
#define NHP_U_SMALL 4   // G <= 8
#endif
#ifndef NHP_U_MID
#define NHP_U_MID 4     // G = 16, 32 (tools/uvar.sh: 4 beats 2 by 5-8 % at K=64, 512)
#endif

struct nhp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *d_results = nullptr;        // [NHP_MAX_SLOTS] log-likelihood results
    double *h_results = nullptr;        // pinned mirror
    double *d_partials = nullptr;       // per-workgroup partial sums
    size_t partials_cap = 0;            // in doubles
    void *d_scratch = nullptr;          // general scratch (gradients, sampler output)
    size_t scratch_cap = 0;             // bytes
    unsigned int *d_counter = nullptr;  // arrival ticket of the fused last-block reduction (kept at 0 between launches)
    // Second lane for independent evaluations inside one call (nhp_cont_loglik_batch): its own stream, partial sums and
    // tickets.  A launch has fixed costs -- dispatch / completion, column staging, the reduction tail -- during which
    // the chip idles; with two lanes they run under the other lane's pair loops (profiles/README.md: two streams).
    // Deferred sampler error (nhp_cont_gibbs_step): the sweep's "weights do not sum to a positive finite value" flag is
    // copied to a pinned word behind the sweep and looked at when the NEXT sweep has been enqueued (or at any call that
    // synchronises), so a chain keeps one step in flight instead of draining the GPU every step.
    int *d_err = nullptr, *h_err = nullptr;
    hipEvent_t ev_err = nullptr;
    bool err_pending = false;
    void *h_stage = nullptr;            // pinned staging buffer for large downloads (gradients, parameters, moments)
    size_t stage_cap = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double *d_partials2 = nullptr;
    size_t partials2_cap = 0;
    unsigned int *d_counter2 = nullptr;
    int cu_count = 256;
    // the device optimizer's state (nhp_lbfgs.h): kept between mle! runs -- allocating and freeing ~400 MB per run cost
    // several ms of a 30 ms run
    void *d_mle = nullptr;
    size_t mle_cap = 0;
    double *h_mle_scal = nullptr;       // pinned scalars of its readbacks
    std::string err;
};

// One contiguous run of bucketed children of a single node, processed by one workgroup.
struct nhp_item {
    int32_t node;      // child node c (0-based)
    int32_t kbeg;      // first child slot in bucket order
    int32_t kend;      // one past the last
    int32_t first;     // bit 0: the first item of its node (owns the column's integral term); bit 1: its only item
};

// A child event in node-bucketed order: everything a wave needs to walk its window.
struct __attribute__((aligned(16))) nhp_child {
    double t;          // event time
    int32_t first;     // index of the first parent inside the look-back window
    int32_t idx;       // original (time-ordered) index of this event; window = [first, idx)
};

// An event in time order, packed so that one 16-byte load fetches a parent.
struct __attribute__((aligned(16))) nhp_event {
    double t;
    int32_t node;      // 0-based
    int32_t pad;
};

struct nhp_cont_dataset {
    nhp_ctx *ctx = nullptr;
    uint64_t uid = 0;                   // process-unique id (caches keyed on "this dataset", never on its address or size)
    int64_t M = 0;
    int32_t N = 0;
    double duration = 0.0, dt_max = 0.0, t_last = 0.0;
    int64_t pairs = 0;
    int32_t group = 8;                  // lanes cooperating on one child in the windowed kernels
    int32_t n_items = 0;
    int32_t max_item = 0;               // most children in one item (sizes the deferred-log LDS buffer)
    int32_t max_window = 0;             // longest look-back window, in parents
    nhp_event *d_ev = nullptr;          // [M] time order, packed (t, node)
    // the same records in 8 bytes: node << 48 | round((t - ev8_t0) * ev8_scale), ev8_scale = 2^s with the largest s that keeps
    // the time field below 2^48 (a resolution of 2^-49 of the data's span).  Read by the short-window log-likelihood kernel,
    // whose time goes into fetching ~8 scattered parent records per event: half the bytes per record.
    uint64_t *d_ev8 = nullptr;
    double ev8_t0 = 0.0, ev8_scale = 0.0;
    // Short windows only (pairs <= 40 per event on average, finite dt_max): the parent-child pairs themselves, child by child
    // in child_w order, most recent parent first -- node << 48 | Δt as a 48-bit fraction of dt_max, 8 bytes a pair.  The
    // exponential log-likelihood kernel then STREAMS its item's pairs (contiguous) instead of fetching ~8 scattered records per
    // child behind each child record.  d_poff: pair offsets by child_w position (host-made with the dataset);
    // d_plist: built on the device at the first evaluation that wants it.
    uint32_t *d_poff = nullptr;         // [M + 1]
    uint64_t *d_plist = nullptr;        // [pairs]
    // the same pairs for the logit-normal parent sampler: {logit(x), 1/(x(1-x))} at x = Δt/Δtmax -- the data half of the
    // impulse pdf, evaluated ONCE with the operation sequence the sampler (and the oracle) use, so the weights keep their
    // bits -- and the parent's node; built at the first logit-normal sweep
    double2 *d_plq = nullptr;           // [pairs]
    uint16_t *d_pnode = nullptr;        // [pairs]
    // Child slices (cont_slices.hip): the same short-window pairs, laid out for one LANE per child.  A slice is
    // 64 consecutive children of one item in child_w order = one wavefront's work; its pairs are stored row by row, row r
    // = the r-th most recent parent of each of the 64 children (64 records, one per lane: a wave-wide coalesced load), as
    // many rows as the slice's longest window (children are sorted by window length, so a slice's windows are nearly
    // equal; shorter ones are padded with zero-weight records on node N).  A record is 6 bytes in two planes: lo = low 32
    // bits of the delay, hi = node << (16 - nb) | high bits of the delay; the delay is round(Δt/Δtmax · 2^(48 - nb)),
    // nb = bit length of N.  d_sl_row / d_sl_item0 are host-made with the dataset; the planes are filled on the device at
    // the first evaluation that uses them.
    uint32_t *d_sl_row = nullptr;       // [n_slices + 1] first row of each slice
    int32_t *d_sl_item0 = nullptr;      // [n_items + 1] first slice of each item
    uint32_t *d_sl_lo = nullptr;        // [(sl_rows + 16) * 64]
    uint16_t *d_sl_hi = nullptr;        // [(sl_rows + 16) * 64]
    // logit-normal impulses: the data half of every record's pdf, {logit(x), 1/(x(1-x))} at x = Δt/Δtmax from the EXACT times with
    // the operation sequence of nhp_logitnormal_data (the bits of d_plq), in the same rows as d_sl_lo / d_sl_hi (whose node
    // field the consumers read); padding records hold {0, 0}.  Built at the first logit-normal parent sweep over the slices.
    double *d_sl_L = nullptr, *d_sl_Q = nullptr;   // [(sl_rows + 16) * 64] each
    double *d_sl_D = nullptr;                       // [(sl_rows + 16) * 64] exact delays t_i - t_j (exponential parent sampler)
    int64_t sl_rows = 0;
    int32_t n_slices = 0, sl_nb = 0;    // node bits
    int32_t sl_max_rows = 0;            // most rows of one slice
    // Parent slices (cont_slices.hip, the gradient's second phase): the same pairs of every item grouped by PARENT node --
    // lane = parent node (the item's parent nodes sorted by their number of pairs, 64 to a slice, every node present),
    // row r = the node's r-th pair as {child slot inside the item, delay}; a record is again 6 bytes: lo = low 32 bits of
    // the delay, hi = slot << (16 - sb) | high bits, sb = bit length of max_item; padding records point at slot max_item
    // (whose 1/λ is 0).  Built on the device at the first gradient that uses them; every lane's records are sorted by
    // (slot, delay), so the list -- and with it the order of every gradient sum -- is the same for every build.
    uint32_t *d_ps_row = nullptr;       // [n_items * ps_spi + 1] first row of each parent slice
    uint16_t *d_ps_perm = nullptr;      // [n_items * ps_spi * 64] parent node of each lane (0xFFFF: none)
    uint32_t *d_ps_lo = nullptr;        // [(ps_rows + 16) * 64]
    uint16_t *d_ps_hi = nullptr;
    int64_t ps_rows = 0;
    int32_t ps_spi = 0, ps_sb = 0;      // parent slices per item = ceil(N / 64); slot bits
    bool all_sole = false;              // every item is the only item of its node (the gradient then stores, never adds)
    // device arrays
    double *d_times = nullptr;          // [M] time order
    int32_t *d_nodes = nullptr;         // [M] 0-based
    nhp_child *d_child = nullptr;       // [M] bucket order (time order inside a node)
    // column shard (nhp_cont_dataset_create_columns): only children on nodes [col_begin, col_end) are evaluated, every
    // event is still a parent; the whole dataset has col_begin = 0, col_end = N
    int32_t col_begin = 0, col_end = 0;
    nhp_child *d_child_w = nullptr;
    int32_t *d_wpos = nullptr;          // [M] bucket position (index into `child`) of the child at each child_w position     // [M] same, but inside each item sorted by window length (windowed kernels)
    int32_t *d_boff = nullptr;          // [N+1] bucket offsets
    nhp_item *d_items = nullptr;        // [n_items]
    double *d_cnt = nullptr;            // [N] events per node
    int32_t *d_pn = nullptr;            // [M] bucket order: parent node of each child from the latest parent sweep (-1 = baseline)
    mutable bool pn_valid = false;      // d_pn holds an assignment (set by the sampler / nhp_cont_lgcp_loglik)
    // adjacency sweep: the data-only pair lists (per child column, sorted by parent node), built on first use
    int32_t *d_adj_k = nullptr, *d_adj_p = nullptr;   // [pairs] child slot within the column, parent node
    double *d_adj_dt = nullptr;                       // [pairs] t_child - t_parent
    double2 *d_adj_lq = nullptr;                      // [pairs] logit-normal impulses: {logit(x), 1/(x(1-x))} at x = Δt/Δtmax (data only; {0, 0} outside (0, 1)), made at the first logit-normal sweep
    int32_t *d_adj_start = nullptr;                   // [N*(N+1)] per-column offsets by parent node
    int64_t *d_adj_off = nullptr;                     // [N+1] first pair of each column
    // recursive ll evaluated as a truncated window (cont_recursive.hip): children with window starts for `cut_cached`
    nhp_child *d_child_cut = nullptr;
    mutable double cut_cached = -1.0;
    mutable int64_t cut_pairs = 0;
    int64_t n_zero_time = 0;            // events at exactly t = 0.0 (the recursion's seen-flag skips them: SURVEY D9)
    // O(M·N) recursion, one wave per (column, part of the parent nodes) (k_recursive_waves): the events with t > 0 part by
    // part (rec_h parts of rec_np nodes; node = index inside the part; 192 records of padding), the parts' offsets and,
    // per part and bucket position, how many of the part's events precede the child.  Data only; made at the first call.
    nhp_event *d_rec_ev = nullptr;
    int32_t *d_rec_poff = nullptr;      // [rec_h + 1]
    int32_t *d_rec_rank = nullptr;      // [rec_h][M]
    int32_t rec_np = 0, rec_h = 0;
    unsigned char *d_adj_group = nullptr;             // [N*N] grouping code of entry (p, c): see k_adj_build
    // host copies kept for host-side helpers
    std::vector<int32_t> h_boff;
    std::vector<double> h_cnt;
    // crowding of the data: h_slab_max[k] = most events inside any closed time window of length h_slab_len[k]
    // (lengths double from the mean gap up to the duration).  Tightens the truncated-window bound of the recursive
    // formulation (cont_recursive.hip): events older than `cut` arrive at most h_slab_max per slab.
    std::vector<double> h_slab_len;
    std::vector<int64_t> h_slab_max;
    bool slab_done = false;             // the two vectors above have been made (nhp_dataset_slab_stats: at the recursion's first bound)
    std::vector<int64_t> h_pair_off;    // [N+1] prefix of window pairs per child node (adjacency sweep scratch)
};

struct nhp_cont_model {
    nhp_ctx *ctx = nullptr;
    int32_t N = 0, baseline_kind = 0, grid_n = 0, impulse_kind = 0, has_A = 0;
    double dt_max = 0.0, grid_end = 0.0;
    double *d_lambda0 = nullptr, *d_grid = nullptr;
    double *d_p1 = nullptr;             // theta (exp) or mu (logit-normal)
    double *d_p2 = nullptr;             // tau (logit-normal)
    double *d_W = nullptr, *d_A = nullptr;
    // parameters change through create/update/set_params and the device-side draws: `version` counts the changes, so
    // quantities derived from the parameters (the recursive path's truncation window) are recomputed only when stale
    uint64_t version = 1;
    mutable uint64_t rec_version = 0;
    mutable uint64_t rec_ds = 0;        // uid of the dataset whose slab statistics the cached bound was derived from
    mutable double rec_cut = 0.0;       // look-back beyond which the full-history sum is below 2^-60 of every λ_i (0: no bound)
    // running sums of the chain's samples (nhp_cont_model_moments_*): Σx and Σx² over [λ0; θ | μ; τ; W; vec(A)], so a
    // chain's posterior summaries never cross PCIe step by step
    double *d_mom = nullptr;            // [2][mom_len]
    int64_t mom_len = 0, mom_count = 0;
    // BernoulliNetworkModel.ρ on the device (nhp_cont_model_set_rho / nhp_cont_network_step): {ρ, Σρ, Σρ², ΣA of the
    // latest sweep}; the sums follow the moments above
    double *d_rho = nullptr;
};

// RCCL communicator of one rank (comm.hip); librccl.so.1 is dlopen'ed on first use
struct nhp_comm {
    nhp_ctx *ctx = nullptr;
    void *nccl = nullptr;               // ncclComm_t
    int rank = 0, world = 1;
};
// in-place sum / gather of device doubles over the ranks, on the ctx stream (asynchronous)
nhp_status nhp_comm_allreduce_dev(nhp_ctx *ctx, nhp_comm *comm, double *d_buf, size_t n);
nhp_status nhp_comm_allgather_dev(nhp_ctx *ctx, nhp_comm *comm, const double *d_mine, size_t n, double *d_all);
// cont_grad.hip: enqueue log-likelihood (-> ctx->d_results[0]) + gradient (-> *d_grad, P doubles in ctx->d_scratch, with one
// spare double in front of it at (*d_grad)[-1] for the packed [ll; grad] exchange)
nhp_status nhp_grad_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags, int64_t grad_len,
                            double **d_grad);
// comm.hip: the same, with the shards' [ll; grad] summed over the ranks on the device when `comm` is given (nullable)
nhp_status nhp_grad_enqueue_reduced(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags,
                                    int64_t grad_len, double **d_grad);
// cont_adjacency.hip: enqueue one sweep of A; per-column link counts land at *d_links [N] (scratch)
nhp_status nhp_adj_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m, const double *rho_matrix, double rho,
                           const double *d_rho_scalar, const double *u, uint64_t seed, uint64_t step, double **d_links);


// Kernel-side view of model + data (passed by value).
struct nhp_cont_args {
    const double *times;
    const int32_t *nodes;
    const nhp_event *ev;
    const uint64_t *ev8;             // packed records (or null): see nhp_cont_dataset::d_ev8
    const uint32_t *poff;            // pair list of short-window datasets (or null): see nhp_cont_dataset::d_poff
    const uint64_t *plist;
    const double2 *plq;              // logit-normal pair cache (or null): see nhp_cont_dataset::d_plq
    const uint16_t *pnode;
    double ev8_t0, ev8_scale, ev8_inv;
    const nhp_child *child;
    const nhp_child *child_w;
    const int32_t *wpos;             // child_w position -> bucket position
    int32_t col_begin, col_end;      // columns (child nodes) this dataset owns
    const int32_t *boff;
    const nhp_item *items;
    const double *cnt;
    const double *lambda0, *grid;
    const double *p1, *p2, *W, *A;
    int64_t M;
    int32_t N, grid_n, baseline_kind, impulse_kind;
    double dt_max, inv_dtmax, duration;
    int32_t dbg;        // phase-ablation bits; only read in -DNHP_ABLATE builds (tools/ablate.sh)
};

struct nhp_disc_dataset {
    nhp_ctx *ctx = nullptr;
    int32_t N = 0, B = 0, L = 0;
    int64_t T = 0;
    double *d_dataT = nullptr;          // [T*N] counts as f64, t fastest (GEMM operand)
    uint8_t *d_data8 = nullptr;         // [T*N] the same counts in one byte each (when none exceeds 255): what the convolution reads
    double *d_conv = nullptr;           // [T*N*B] t fastest
    double *d_colsum = nullptr;         // [2N] Σ_t data[n,t], then Σ_t loggamma(data[n,t]+1)
    double lgamma_sum = 0.0;            // Σ_{n,t} loggamma(data[n,t]+1): the data-only term of the Poisson ll
    // occupied bins (count > 0) in time-major order, for the discrete adjacency sweep
    int64_t nocc = 0;                                 // occupied bins
    // the discrete adjacency sweep's entry lists: the time axis cut into `da_nspans` spans of at most NHP_DA_SPAN bins (a
    // balanced round of two workgroups per CU where T allows); within a span the occupied bins are sorted by (node, bin)
    // and padded with empty entries (count 0) to a multiple of 4, so a thread takes whole 16-byte groups of consecutive
    // entries that mostly share their node
    int32_t da_nspans = 0, da_max_entries = 0;        // spans; the most entries (incl. padding) any span holds
    int64_t nocc_pad = 0;                             // entries incl. padding
    int32_t *d_occ_t = nullptr, *d_occ_c = nullptr;   // [nocc_pad] bin, node (0-based)
    double *d_occ_s = nullptr;                        // [nocc_pad] the count
    uint32_t *d_occ_pack = nullptr;                   // [nocc_pad] node << 16 | count << 8 | bin % 256 (null when a node index or count does not fit)
    int32_t *d_occ_off = nullptr;                     // [da_nspans + 1] first entry of each span
    int32_t *d_span_t = nullptr;                      // [da_nspans + 1] first bin of each span
    double *d_convsum = nullptr;                      // [N*B] Σ_t Ŝ[t, p, b] (filled with the convolution)
    // time-varying baseline of a DiscreteLogGaussianCoxProcess (nhp_disc_set_lgcp_baseline); used when a call passes lambda0 = NULL
    double *d_baseT = nullptr;                        // [T*N] baseline intensity per bin, t fastest
    int32_t *d_base_counts = nullptr;                 // [T*N] events attributed to the baseline by the latest parent sweep
    bool base_counts_valid = false;
    std::vector<double> h_grid_x;                     // the LGCP grid [G]
};

#ifndef NHP_DA_SPAN
#define NHP_DA_SPAN 256  // bins of a workgroup's span in the discrete adjacency sweep, at most (the packed entry keeps bin % 256)
#endif

// disc.hip pieces shared with disc_gibbs.hip: upload W, θ, A (and λ0) and build the bump table E [N·B x N] on the device
// (GEMM order k = p + b·N, or the reference's category order p·B + b with cat_order) plus base[c] = λ0[c]·dt; `extra`
// doubles of scratch follow at *extra_ptr.  Layout after E: base (N) | λ0 (N) | W (N²) | θ (N²B) | A (N²) | extra.
nhp_status nhp_disc_stage_bump(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0, const double *W,
                               const double *theta, const double *A, double dt, double **E, double **base, size_t extra,
                               double **extra_ptr, int cat_order = 0);
nhp_status nhp_disc_launch_intensity(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *E, const double *base,
                                     bool per_bin_baseline, double *dlam);

// ---- error plumbing -------------------------------------------------------------------
void nhp_set_error(nhp_ctx *ctx, const char *fmt, ...);
#define NHP_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            nhp_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NHP_EHIP;                                                                 \
        }                                                                                    \
    } while (0)
#define NHP_TRY(expr)                      \
    do {                                   \
        nhp_status s_ = (expr);            \
        if (s_ != NHP_OK) return s_;       \
    } while (0)

nhp_status nhp_ctx_reserve_partials(nhp_ctx *ctx, size_t n_doubles);
nhp_status nhp_dataset_slab_stats(nhp_ctx *ctx, const nhp_cont_dataset *ds);      // cont_data.hip
int nhp_pick_group(double mean_window);
// recursion_cost: what the caller's O(M·N) kernel costs relative to its windowed route, in units of the log-likelihood's
// ratio (1 for the log-likelihood; the gradient's recursion is 3.5x the log-likelihood's, its windowed route 2.9x)
nhp_status nhp_recursive_window(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child **child_cut,
                                int *group, double recursion_cost = 1.0);
nhp_status nhp_launch_recursive_flags(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags, double *d_out);
nhp_status nhp_launch_event_intensity_as(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child *child_w,
                                         int group, int mask_integral, double *d_lambda);
nhp_status nhp_launch_windowed_as(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child *child_w,
                                  int group, int mask_integral, double *d_out);
nhp_status nhp_ctx_reserve_scratch(nhp_ctx *ctx, size_t bytes);
nhp_status nhp_ensure_pair_cache(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_args *a);   // cont_sampler.hip
// cont_slices.hip: the exponential log-likelihood of the dataset's own short windows, one lane per child over the child
// slices.  *launched = false (and NHP_OK) when the dataset has no slices or the model is not covered.
nhp_status nhp_launch_windowed_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int mask_integral,
                                      double *d_out, bool *launched);
nhp_status nhp_launch_windowed_slices_ln(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int mask_integral,
                                      double *d_out, double *d_lambda, bool *launched);
// the same with the analytic gradient (params! order, P doubles at d_grad) from one fused launch over the child and the
// parent slices; *launched = false when the pair is not covered
nhp_status nhp_launch_grad_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out, double *d_grad,
                                  bool *launched);
// true: that launch stores every entry of the gradient itself; false: it adds to what k_grad_init left
bool nhp_grad_slices_direct(const nhp_cont_dataset *ds, const nhp_cont_model *m);
// resample_parents for logit-normal impulses over the child slices (one lane per child, rows coalesced, the first weights of a
// child kept in LDS between the sum and the scan): same bits as k_sampler.  *launched = false when not covered.
nhp_status nhp_launch_sampler_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const double *d_u, uint64_t seed,
                                     uint64_t step, int64_t *parents, int64_t *pnodes, int32_t *pn_b, double *dt_b, int *d_err, bool *launched);
// S = 2 or 4 exponential models with homogeneous baselines on one dataset in ONE pass over its child slices (every record is
// fetched and decoded once, S columns sit in LDS): results -> ctx->d_results[slot0 .. slot0 + S).  *launched = false when the
// models are not covered (the caller falls back to its other kernels).
nhp_status nhp_launch_slices_batch(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *const *ms, int S, int32_t slot0,
                                   bool *launched);
nhp_cont_args nhp_make_args(const nhp_cont_dataset *ds, const nhp_cont_model *m);
nhp_status nhp_check_pair(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m);
// Device -> caller memory through the context's pinned staging buffer: DMA at link speed into pinned memory, then one
// host copy (a direct copy into pageable caller memory ran at ~1 GB/s here).  Synchronises the stream.
nhp_status nhp_download(nhp_ctx *ctx, void *dst, const void *d_src, size_t bytes);
// report (once) the deferred sampler error of an earlier nhp_cont_gibbs_step, waiting for that sweep if need be
nhp_status nhp_check_deferred(nhp_ctx *ctx);
bool nhp_is_column_shard(const nhp_cont_dataset *ds);
// entry points that need every column (samplers, intensity tables): refuse a column shard
#define NHP_WHOLE_DATASET(ctx, ds, what)                                                                       \
    do {                                                                                                      \
        if (nhp_is_column_shard(ds)) {                                                                        \
            nhp_set_error(ctx, what ": not available on a column shard (log-likelihood and gradient only)");  \
            return NHP_ENOTIMPL;                                                                              \
        }                                                                                                     \
    } while (0)

// launchers implemented in the kernel translation units (all asynchronous on ctx->stream)
nhp_status nhp_launch_windowed(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out);
nhp_status nhp_launch_recursive(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out);
nhp_status nhp_launch_finalize(nhp_ctx *ctx, const nhp_cont_args &a, int n_partials, double *d_out);
nhp_status nhp_launch_event_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_lambda);

// wave-local ordering of LDS traffic: wait for this wave's outstanding LDS operations (lgkmcnt(0)
// only -- global stores stay in flight) and keep the compiler from moving LDS accesses across it
#define NHP_LDS_SYNC()                                   \
    do {                                                 \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_s_waitcnt(0xc07f);              \
        __builtin_amdgcn_wave_barrier();                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)


// ---- wave-partitioned O(M·N) recursion (cont_recursive.hip; its gradient pass in cont_grad.hip) -----------------------
struct nhp_rec_parts {          // kernel-side view of the per-part lists (nhp_cont_dataset::d_rec_*)
    const nhp_event *ev;        // [Σ part lengths + 192] events with t > 0 (D9), part by part, time order inside; node = index inside the part
    const int32_t *poff;        // [H + 1] first record of each part
    const int32_t *rank;        // [H][M] by bucket position k: events of the part with time index < idx_k
};
nhp_status nhp_rec_parts_for(nhp_ctx *ctx, const nhp_cont_dataset *ds, int *PQ, int *H, nhp_rec_parts *rp);
nhp_status nhp_launch_recursive_waves(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out, double *d_ginv,
                                      bool *launched);
// the (parents per lane, parts) shapes both kernels are instantiated for: X(PQ, H) launches, `otherwise` runs when none fits
#define NHP_REC_SHAPES(X, otherwise)              \
    do {                                          \
        if (PQ == 1 && H == 1) X(1, 1);           \
        else if (PQ == 1 && H == 2) X(1, 2);      \
        else if (PQ == 1 && H == 4) X(1, 4);      \
        else if (PQ == 2 && H == 4) X(2, 4);      \
        else if (PQ == 2 && H == 8) X(2, 8);      \
        else if (PQ == 4 && H == 4) X(4, 4);      \
        else if (PQ == 4 && H == 8) X(4, 8);      \
        else if (PQ == 4 && H == 16) X(4, 16);    \
        else if (PQ == 8 && H == 2) X(8, 2);      \
        else { otherwise; }                       \
    } while (0)
// Ordering of one wave's own LDS traffic without draining it: the LDS unit executes a wave's instructions in issue order (an
// atomic of lane A is seen by a later read of lane B of the same wave), so only the compiler must be kept from reordering.
#define NHP_LDS_ORDER()                                         \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#ifndef NHP_RECW_SYNC
#define NHP_RECW_SYNC() NHP_LDS_ORDER()
#endif

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ double nhp_dpp_add(double v, const int sel)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int plo, phi;
    switch (sel) {
    case 0: plo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); phi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: plo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); phi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: plo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); phi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: plo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); phi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return v + __hiloint2double(phi, plo);
}

// Sum over the 64 lanes of a wave; every lane returns the total.  Rows of 16 are reduced in the
// VALU with DPP, the four row sums are combined through scalar lane reads (no LDS round trips).
__device__ __forceinline__ double nhp_wave_sum(double v)
{
    v = nhp_dpp_add(v, 0);
    v = nhp_dpp_add(v, 1);
    v = nhp_dpp_add(v, 2);
    v = nhp_dpp_add(v, 3);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    r += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    r += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    r += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return r;
}

// Maximum over the 64 lanes of a wave, in the VALU/SALU only (DPP inside rows of 16, scalar reads across rows): no LDS round trips.
__device__ __forceinline__ int nhp_wave_max_i32(int v)
{
    int o;
    o = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;      // quad_perm [1,0,3,2]
    o = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;      // quad_perm [2,3,0,1]
    o = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); v = o > v ? o : v;     // row_half_mirror
    o = __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); v = o > v ? o : v;     // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

__device__ __forceinline__ double nhp_wave_sum_shfl(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over a block of NW waves; result valid in thread 0.  `red` holds >= NW doubles.
template <int NW>
__device__ __forceinline__ double nhp_block_sum_n(double v, double *red)
{
    v = nhp_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ double nhp_block_sum(double v, double *red) { return nhp_block_sum_n<NHP_WAVES>(v, red); }

// Two sums with one pair of barriers; results valid in thread 0.  `red` holds >= 2·NW doubles.
template <int NW>
__device__ __forceinline__ void nhp_block_sum2_n(double &x, double &y, double *red)
{
    x = nhp_wave_sum(x);
    y = nhp_wave_sum(y);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = x; red[NW + (threadIdx.x >> 6)] = y; }
    __syncthreads();
    double sx = 0.0, sy = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < NW; ++w) { sx += red[w]; sy += red[NW + w]; }
    x = sx; y = sy;
}

// Phase ablation for timing experiments (tools/ablate.sh): compiled in only with -DNHP_ABLATE,
// where env NHP_DBG selects phases to skip (results are then wrong by design).
#ifdef NHP_ABLATE
#define NHP_SKIP(a, bit) ((a).dbg & (bit))
#else
#define NHP_SKIP(a, bit) 0
#endif

// workgroup size of the windowed log-likelihood kernel (its own knob: more waves per staged column)
#ifndef NHP_WBLOCK
#define NHP_WBLOCK 256
#endif
