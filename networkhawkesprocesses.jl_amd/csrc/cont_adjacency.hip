// Gibbs sweep of the adjacency matrix (reference: resample_adjacency_matrix! / resample_column!
// src/continuous.jl:444-487, integrated_intensity :489-498, sum_log_intensity :500-519,
// logsumexp src/utils/helpers.jl:13-16, link_probability src/networks.jl:65-68).
//
// The reference evaluates, for each of the N² entries, two full passes over every event of the child
// node (2·N² event scans per sweep).  Columns are independent; inside column c the entries are
// sampled one parent node after the other, each conditional on the current state of the column.
//
// Here column c is owned by one workgroup per phase.  Phase 1 (k_adj_pairs, 256 lanes) evaluates every
// parent-child pair of the column ONCE, x = W[p,c]·ħ(Δt), and counting-sorts the pairs by parent node
// p into (child slot, x) lists, while λ_k = λ0 + Σ A[p,c]·x is accumulated per child.  Phase 2
// (k_adj_sweep, one wave) walks p = 1..N: only the children that have a parent on node p are touched --
//     ll1 - ll0 = -W[p,c]·cnt[p] + Σ_k [log(λ_k⁰ + x_kp) - log λ_k⁰] + log ρ - log(1-ρ),
// (the baseline integral and the first event's dropped term, SURVEY D10, cancel) -- the entry is
// drawn with the reference's Bernoulli rule u <= exp(ll1 - logsumexp(ll0, ll1)), and the affected
// λ_k are updated.  Work per sweep: Σ pairs instead of 2·N²·M.  The phases are separate kernels because
// the sweep is a serial chain per column: what matters is that ALL columns run at once, and a one-wave
// workgroup with 20 bytes of LDS per child fits 6+ to a CU where the fused kernel (72 KB) fit two.
#include <algorithm>

#include "nhp_internal.h"
#include "nhp_math.h"

#define NHP_ADJ_RUN_SHIFT 24      // parent word of a cached entry: node | (repeats folded into the entry) << 24

__device__ __forceinline__ double adj_baseline(const nhp_cont_args &a, int c, double t)
{
#pragma clang fp contract(off)
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return a.lambda0[c];
    const double *x = a.grid;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    int lo = 0, hi = a.grid_n - 1;
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

// The pair lists depend on the DATA only -- which (parent, child) pairs exist, their delays, the parent's
// node -- so they are built once per dataset (k_adj_build) and kept: per column, entries sorted by parent
// node, each {child slot, parent node, Δt}, plus the per-column offsets by parent node.
__global__ __launch_bounds__(NHP_BLOCK) void k_adj_build(nhp_cont_args a, const int64_t *__restrict__ pair_off, int group,
                                                         int32_t *__restrict__ ent_k, int32_t *__restrict__ ent_p,
                                                         double *__restrict__ ent_dt, int32_t *__restrict__ col_start,
                                                         unsigned char *__restrict__ col_group, int max_children)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, c = a.col_begin + blockIdx.x, tid = threadIdx.x;
    // `group` lanes share a child (the dataset's lanes-per-child width, a power of two <= 64), so a wave
    // has 64/group windows in flight instead of one
    const int gl = tid & (group - 1), gid = tid / group, ngroups = NHP_BLOCK / group;
    int *start = reinterpret_cast<int *>(smem);                            // [N + 1] pair-list offsets by p
    int *cursor = start + N + 1;                                           // [N]
    int *scan_tmp = cursor + N;                                            // [NHP_BLOCK]
    const int kb = a.boff[c], ke = a.boff[c + 1], nchild = ke - kb;
    for (int p = tid; p < N; p += NHP_BLOCK) cursor[p] = 0;
    __syncthreads();
    // ---- pairs per parent node
    for (int k = gid; k < nchild; k += ngroups) {
        const nhp_child ch = a.child[kb + k];
        for (int j = ch.first + gl; j < ch.idx; j += group) atomicAdd(&cursor[a.nodes[j]], 1);
    }
    __syncthreads();
    // exclusive scan of cursor -> start (each thread scans a contiguous chunk, thread 0 the chunk sums)
    const int chunk = (N + NHP_BLOCK - 1) / NHP_BLOCK;
    int local = 0;
    for (int q = 0; q < chunk; ++q) { const int p = tid * chunk + q; if (p < N) local += cursor[p]; }
    scan_tmp[tid] = local;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < NHP_BLOCK; ++t) { const int v = scan_tmp[t]; scan_tmp[t] = run; run += v; }
        start[N] = run;
    }
    __syncthreads();
    {
        int run = scan_tmp[tid];
        for (int q = 0; q < chunk; ++q) {
            const int p = tid * chunk + q;
            if (p < N) { const int v = cursor[p]; start[p] = run; run += v; }
        }
    }
    __syncthreads();
    for (int p = tid; p < N; p += NHP_BLOCK) cursor[p] = 0;
    __syncthreads();
    // ---- scatter by parent node
    const int64_t base = pair_off[c];
    for (int k = gid; k < nchild; k += ngroups) {
        const nhp_child ch = a.child[kb + k];
        for (int j = ch.first + gl; j < ch.idx; j += group) {
            const nhp_event e = a.ev[j];
            const int pos = start[e.node] + atomicAdd(&cursor[e.node], 1);
            ent_k[base + pos] = k;
            ent_p[base + pos] = e.node;
            ent_dt[base + pos] = ch.t - e.t;
        }
    }
    for (int p = tid; p <= N; p += NHP_BLOCK) col_start[(size_t)c * (N + 1) + p] = start[p];
    // ---- a child named more than once by one parent node (two events of p inside the child's window): the entries of a short
    // list are put in (child, time) order and the repeats folded into the first of their run -- it carries the run length in the
    // high bits of its parent word, k_adj_eval sums the run's x into it in that order, the repeats become dead entries
    // (child -1).  Every short list then names each child once and is eligible for the grouped step; only lists longer than
    // 16 entries take the general path (they were ~10 % of the steps of the 1024-node case and a quarter of its time).
    __syncthreads();
    __threadfence_block();
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const int eb = start[p], n = start[p + 1] - eb;
        if (n < 2 || n > 16) continue;
        int kk[16];
        double dd[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { kk[i] = i < n ? ent_k[base + eb + i] : 0x7fffffff; dd[i] = i < n ? ent_dt[base + eb + i] : 0.0; }
        // odd-even transposition sort on (k ascending, Δt descending = parent time ascending): fixed network, registers only
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = r & 1; i + 1 < 16; i += 2) {
                const bool swap = kk[i] > kk[i + 1] || (kk[i] == kk[i + 1] && dd[i] < dd[i + 1]);
                const int k0 = kk[i], k1 = kk[i + 1];
                const double d0 = dd[i], d1 = dd[i + 1];
                kk[i] = swap ? k1 : k0; kk[i + 1] = swap ? k0 : k1;
                dd[i] = swap ? d1 : d0; dd[i + 1] = swap ? d0 : d1;
            }
        }
        int run = 0;                                      // repeats that follow entry i (walking backwards)
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            if (i >= n) continue;
            run = (i + 1 < n && i + 1 < 16 && kk[i + 1 < 16 ? i + 1 : 15] == kk[i]) ? run + 1 : 0;
            const bool rep = i > 0 && kk[i > 0 ? i - 1 : 0] == kk[i];
            ent_k[base + eb + i] = rep ? -1 : kk[i];
            ent_dt[base + eb + i] = dd[i];
            ent_p[base + eb + i] = p | ((rep ? 0 : run) << NHP_ADJ_RUN_SHIFT);
        }
    }
    // Grouping (data only).  Entries (p, c) and (p', c) of a column interact only through children that have
    // parents on both nodes, so consecutive parents whose lists share no child can be decided together with
    // the same result as one after the other.  col_group[p + c·N]: 1..4 = p heads a group of that many
    // consecutive parents -- each list at most 16 entries, all children distinct inside the group, all in
    // one 64-parent chunk -- which the sweep decides in ONE step (16 lanes per parent); 0 = member of the
    // group before it; 255 = p is decided alone by the general path (long list, or a child named twice).
    __syncthreads();
    __threadfence_block();
    if (tid == 0) {
        int *seen = scan_tmp + NHP_BLOCK;                // [max_children] group id that last touched the child
        for (int k = 0; k < nchild; ++k) seen[k] = 0;
        int gid = 0, p = 0;
        unsigned char *grp = col_group + (size_t)c * N;
        // list of q: <= 16 entries, every child unseen in group id -- in the group so far AND earlier in q's own list (a child
        // named twice by q goes to the general path: the full-size parity test caught a grouped member summing its two
        // entries apart).  Marks as it goes and takes the marks back on failure, so it is linear in the list.
        auto fits = [&](int q, int id) {
            const int eb = start[q], ee = start[q + 1];
            if (ee - eb > 16) return false;
            for (int e = eb; e < ee; ++e) {
                const int ke = ent_k[base + e];
                if (ke < 0) continue;                    // (a folded repeat)
                int &sk = seen[ke];
                if (sk == id) {
                    for (int e2 = eb; e2 < e; ++e2) if (ent_k[base + e2] >= 0) seen[ent_k[base + e2]] = 0;
                    return false;
                }
                sk = id;
            }
            return true;
        };
        auto take = [&](int, int) {};                    // (fits() has marked the list)
        while (p < N) {
            ++gid;
            const bool ok = start[p + 1] - start[p] <= 16;
            if (ok)                                      // (repeats are folded: a short list names a child once)
                for (int e = start[p]; e < start[p + 1]; ++e)
                    if (ent_k[base + e] >= 0) seen[ent_k[base + e]] = gid;
            if (!ok) { grp[p] = 255; ++p; continue; }
            int g = 1;
            while (g < 4 && p + g < N && ((p + g) & 63) != 0 && fits(p + g, gid)) { take(p + g, gid); grp[p + g] = 0; ++g; }
            grp[p] = (unsigned char)g;
            p += g;
        }
    }
}

__global__ __launch_bounds__(256) void k_adj_lq(int64_t n, double inv_dtmax, const double *__restrict__ ent_dt, double2 *__restrict__ ent_lq)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ent_lq[i] = nhp_logitnormal_data(inv_dtmax, ent_dt[i]);
}

// Per sweep: x = W[p,c]·ħ(Δt) for every cached entry of column c (a streaming pass: coalesced reads of
// {k, p, Δt}, the column of the tables in LDS) and λ_k = λ0 + Σ A[p,c]·x per child.
// TH threads per column: the walk is a stream (24 bytes in, 8 out per entry) behind a column staged once, so a column takes as
// many waves as the CU has room for (512: 76 -> see profiles/README.md)
template <int IMP, int TH>
__global__ __launch_bounds__(TH) void k_adj_eval(nhp_cont_args a, const double *__restrict__ A,
                                                        const int64_t *__restrict__ pair_off,
                                                        const int32_t *__restrict__ ent_k, const int32_t *__restrict__ ent_p,
                                                        const double *__restrict__ ent_dt, const double2 *__restrict__ ent_lq,
                                                        double *__restrict__ ent_x,
                                                        int max_children, double *__restrict__ lam_g,
                                                        const double *__restrict__ rho_mat, double rho_host,
                                                        const double *__restrict__ rho_dev, const double *__restrict__ u,
                                                        uint64_t seed, uint64_t step, double *__restrict__ uni_out,
                                                        double *__restrict__ bias_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, c = a.col_begin + blockIdx.x, tid = threadIdx.x;
    double2 *col = reinterpret_cast<double2 *>(smem);                      // [N] exp {θ, W}; logit {μ, √τ}
    double *colw = reinterpret_cast<double *>(col + N);                    // [N] logit: W
    double *acol = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : N);        // [N] A[·, c]
    double *lam = acol + N;                                                // [max_children] λ_k
    const int kb = a.boff[c], nchild = a.boff[c + 1] - kb;
    for (int p = tid; p < N; p += TH) {
        const size_t k = (size_t)p + (size_t)c * N;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            col[p] = make_double2(a.p1[k], a.W[k]);
        } else {
            col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
            colw[p] = a.W[k];
        }
        acol[p] = A[k];
        // The per-entry constants of the sweep -- logit of the Bernoulli draw, prior log-odds - W·cnt -- are data of this
        // sweep, not of the chain: computed here, 256 lanes wide (the one-wave chain used to spend ~400 dependent
        // instructions per 64 parents on them).  The Bernoulli rule u <= exp(ll1 - logsumexp(ll0, ll1)) = 1/(1 + e^{-d}) is
        // logit(u) <= d (src/continuous.jl:472-487 with [3P] rand(Bernoulli(q)) = rand() <= q).
        const double rho = rho_mat ? rho_mat[k] : (rho_dev ? *rho_dev : rho_host);
        const double uu = u ? u[k] : nhp_philox_uniform(seed ^ 0xBE5466CF34E90C6Cull, step, k);
        uni_out[k] = nhp_log(uu / (1.0 - uu));
        bias_out[k] = -(a.W[k] * a.cnt[p]) + nhp_log(rho) - nhp_log(1.0 - rho);
    }
    for (int k = tid; k < nchild; k += TH) lam[k] = adj_baseline(a, c, a.child[kb + k].t);
    __syncthreads();
    const int64_t e0 = pair_off[c], e1 = pair_off[c + 1];
    // four entries per thread and trip: their loads are in flight together and the four pdf evaluations are independent
    // instruction streams (the one-entry loop exposed one global-load latency per entry)
    constexpr int EU = 4;
    for (int64_t eb = e0 + tid; eb < e1; eb += EU * TH) {
        int p[EU], k[EU];
        double dt[EU], x[EU];
        double2 lq[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int64_t e = eb + u * TH < e1 ? eb + u * TH : eb;       // clamped: value unused
            p[u] = ent_p[e]; k[u] = ent_k[e];
            if (IMP == NHP_IMPULSE_EXPONENTIAL) dt[u] = ent_dt[e];
            else lq[u] = ent_lq[e];                                  // {logit(x), 1/(x(1-x))}: the logarithm and the division are data
        }
        int run[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            run[u] = p[u] >> NHP_ADJ_RUN_SHIFT; p[u] &= (1 << NHP_ADJ_RUN_SHIFT) - 1;
            const double2 q = col[p[u]];
            if (IMP == NHP_IMPULSE_EXPONENTIAL) x[u] = q.y * nhp_pdf_exponential(q.x, dt[u]);
            else x[u] = colw[p[u]] * nhp_pdf_logitnormal_cached(q.x, q.y, lq[u]);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (eb + u * TH < e1) {
                const int64_t e = eb + u * TH;
                // the first entry of a run of repeats carries the run's total, summed in list order (k_adj_build); the
                // repeats themselves (child -1) are dead
                for (int r = 1; r <= run[u]; ++r) {
                    const double2 q = col[p[u]];
                    if (IMP == NHP_IMPULSE_EXPONENTIAL) x[u] += q.y * nhp_pdf_exponential(q.x, ent_dt[e + r]);
                    else x[u] += colw[p[u]] * nhp_pdf_logitnormal_cached(q.x, q.y, ent_lq[e + r]);
                }
                if (k[u] < 0) x[u] = 0.0;
                ent_x[e] = x[u];
                const double av = acol[p[u]];
                if (av != 0.0 && k[u] >= 0) atomicAdd(&lam[k[u]], av * x[u]);
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < nchild; k += TH) lam_g[kb + k] = lam[k];
}

__device__ __forceinline__ float adj_dpp_add_f32(float v, const int sel)
{
    const int b = __float_as_int(v);
    int q;
    switch (sel) {
    case 0: q = __builtin_amdgcn_mov_dpp(b, 0xB1, 0xF, 0xF, true); break;     // quad_perm [1,0,3,2]
    case 1: q = __builtin_amdgcn_mov_dpp(b, 0x4E, 0xF, 0xF, true); break;     // quad_perm [2,3,0,1]
    case 2: q = __builtin_amdgcn_mov_dpp(b, 0x141, 0xF, 0xF, true); break;    // row_half_mirror
    default: q = __builtin_amdgcn_mov_dpp(b, 0x140, 0xF, 0xF, true); break;   // row_mirror
    }
    return v + __int_as_float(q);
}

__device__ __forceinline__ float adj_wave_sum_f32(float v)          // every lane returns the total
{
    v = adj_dpp_add_f32(v, 0); v = adj_dpp_add_f32(v, 1); v = adj_dpp_add_f32(v, 2); v = adj_dpp_add_f32(v, 3);
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

__device__ __forceinline__ float adj_row_sum_f32(float v)           // every lane returns the sum of its row of 16 lanes
{
    v = adj_dpp_add_f32(v, 0); v = adj_dpp_add_f32(v, 1); v = adj_dpp_add_f32(v, 2); v = adj_dpp_add_f32(v, 3);
    return v;
}

__device__ __forceinline__ double adj_readlane(double v, int l)     // l is wave-uniform
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- the sequential Gibbs walk over parent nodes.  A parent node touches only a handful of children,
// so ONE wave walks the column: no block barriers on the 1..N critical path, only wave-local LDS
// ordering (LDS operations of a wave complete in issue order; the fence makes the atomics of all lanes
// visible before any lane reads them back).  The per-entry constants -- logit of the uniform draw,
// prior log-odds - W·cnt, current A -- are computed 64 parents at a time, one per lane, and handed to
// the chain with v_readlane; the entry lists of parent p+1 are fetched while p is processed.
#ifdef NHP_STAMP
__device__ unsigned long long g_adj_stamps[4 * 1024];
extern "C" int nhp_debug_adj_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_adj_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif
__global__ __launch_bounds__(64) void k_adj_sweep(nhp_cont_args a, double *__restrict__ A,
                                                  const int64_t *__restrict__ pair_off,
                                                  const int32_t *__restrict__ ent_k, const double *__restrict__ ent_x,
                                                  const int32_t *__restrict__ col_start, const unsigned char *__restrict__ col_group,
                                                  const double *__restrict__ lam_g,
                                                  const double *__restrict__ uni_g, const double *__restrict__ bias_g,
                                                  int max_children, double *__restrict__ col_links, double *__restrict__ sink)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, c = a.col_begin + blockIdx.x, lane = threadIdx.x;
    double *lam = reinterpret_cast<double *>(smem);                        // [max_children] current λ_k
    double *dx = lam + max_children;                                       // [max_children] Σ x_kp of the current p
    int *marker = reinterpret_cast<int *>(dx + max_children);              // [max_children]
    int *start = marker + max_children;                                    // [N + 2] pair-list offsets by p
    unsigned char *grp = reinterpret_cast<unsigned char *>(start + N + 2); // [N + 1] grouping codes (k_adj_build)
    const int kb = a.boff[c], nchild = a.boff[c + 1] - kb;
    for (int k = lane; k < nchild; k += 64) { lam[k] = lam_g[kb + k]; dx[k] = 0.0; marker[k] = 0; }
    for (int p = lane; p <= N + 1; p += 64) start[p] = col_start[(size_t)c * (N + 1) + (p <= N ? p : N)];
    for (int p = lane; p <= N; p += 64) grp[p] = p < N ? col_group[(size_t)c * N + p] : 0;      // (0: the empty step past the end)
    NHP_LDS_SYNC();

    const int64_t base = pair_off[c];
    int links = 0;
    // Every global access of the walk is issued by ALL lanes, every step, at a clamped address: a load or store under a
    // condition compiles to a skipped block, the number of requests in flight is then unknown to the compiler and its waits
    // become s_waitcnt vmcnt(0) -- a full round trip per step, whatever was prefetched (measured: 1460 -> ... cycles/step).
    const size_t cN = (size_t)c * N;
    const int slot = lane >> 4, sub = lane & 15;
    const int64_t total = pair_off[N];
    const int32_t *ek = ent_k + (base < total ? base : 0);             // (an empty last column reads entry 0, and drops it)
    const double *ex = ent_x + (base < total ? base : 0);
    const double *ug = uni_g + cN, *bg = bias_g + cN;
    double *Ac = A + cN;
    // ---- a group of g <= 4 independent parents p .. p+g-1 (same 64-chunk): 16 lanes per parent, one entry per
    // lane, all children distinct -- one LDS read on the chain, no atomics, no ownership marks.  The fp32
    // screening of the general path applies per parent; if any of them is inside its band all are redone in fp64.
    auto visit_group = [&](const int p, const int g, const int ck, const double cx, const double uni_p, const double bias_p, const double aold) {
        const int pp = p + (slot < g ? slot : 0);
#if defined(ADJ_ABL) && (ADJ_ABL & 2)
        const double lv = 1.0;
#else
        const double lv = ck >= 0 ? lam[ck] : 1.0;
#endif
        const double l0 = ck >= 0 ? lv - aold * cx : 1.0;
        // hardware reciprocal and log2 (≈1 ulp each): the 1e-3 band below is far wider than their error
        float t32 = ck >= 0 ? 0.6931471806f * __builtin_amdgcn_logf(1.0f + (float)cx * __builtin_amdgcn_rcpf((float)l0)) : 0.0f;
        t32 = adj_row_sum_f32(t32);                                  // sum over the parent's 16 lanes
        const double d32 = bias_p + (double)t32;
        const bool sure = fabs(uni_p - d32) > 1e-3 * (1.0 + (double)t32);
        double an = uni_p <= d32 ? 1.0 : 0.0;
        if (__ballot(!sure && slot < g) != 0ull) {                   // wave-uniform: some parent needs the exact sum
            double delta = ck >= 0 ? nhp_log(l0 + cx) - nhp_log(l0) : 0.0;
            delta = nhp_dpp_add(delta, 0); delta = nhp_dpp_add(delta, 1); delta = nhp_dpp_add(delta, 2); delta = nhp_dpp_add(delta, 3);
            an = uni_p <= bias_p + delta ? 1.0 : 0.0;
        }
#if !defined(ADJ_ABL) || !(ADJ_ABL & 4)
        *(slot < g ? &Ac[(unsigned)pp] : &sink[c]) = an;             // (16 lanes, one address, one value)
#endif
        if (slot < g && an != aold && ck >= 0) lam[ck] = lv + (an - aold) * cx;
        // links: one count per parent of the group
        const unsigned long long heads = __ballot((lane & 15) == 0 && slot < g && an == 1.0);
        links += __popcll(heads);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next step reads these λ in program order
    };

    // ---- one parent alone (a long list, or a child named more than once): lane l owns entries eb + l, eb + l + 64, ...
    // Its first 64 entries arrive like a group's, requested one step ahead (ck, cx): nearly every such list is shorter,
    // so the step has no global load on its chain (it had two round trips: ~3 us per step, 1/4 of the sweep).
    auto visit_general = [&](const int p, const int ck, const double cx, const double uni_p, const double bias_p, const double aold) {
        const int eb = start[p], ee = start[p + 1];
        if (ck >= 0) atomicAdd(&dx[ck], cx);
        for (int e = eb + 64 + lane; e < ee; e += 64) atomicAdd(&dx[ent_k[base + e]], ent_x[base + e]);
        NHP_LDS_SYNC();
        double delta = 0.0;
        auto own = [&](const int k) {
            if (atomicExch(&marker[k], p + 1) != p + 1) {           // first entry of child k for this p owns it
                const double d = dx[k];
                const double l0 = lam[k] - aold * d;
                delta += nhp_log(l0 + d) - nhp_log(l0);
            }
        };
        if (ck >= 0) own(ck);
        for (int e = eb + 64 + lane; e < ee; e += 64) own(ent_k[base + e]);
        delta = nhp_wave_sum(delta);
        const double anew = uni_p <= bias_p + delta ? 1.0 : 0.0;    // ll1 - ll0 = bias + delta
        Ac[(unsigned)p] = anew;
        links += anew == 1.0 ? 1 : 0;
        auto settle = [&](const int k) {
            // the first taker carries the child's total
            const double dd = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long *>(&dx[k]), 0ull));
            if (dd != 0.0 && anew != aold) lam[k] += (anew - aold) * dd;
        };
        if (ck >= 0) settle(ck);
        for (int e = eb + 64 + lane; e < ee; e += 64) settle(ent_k[base + e]);
        NHP_LDS_SYNC();
    };

    // entries of the step headed by p, one per lane: a group (slot l/16 -> parent p + slot, entry start + l%16) or the
    // first 64 entries of a lone parent.  bounds() issues the LDS reads of the list bounds BEFORE the current step's work --
    // unconditionally, whatever the step's code, so that they do not wait for the code read issued with them -- and request()
    // turns them into entry indices and sends the global loads AFTER it: neither the LDS latency nor the global round trip
    // sits on the chain.  The constants of the step's parents -- logit of the draw, prior log-odds - W·cnt (k_adj_eval) and
    // the current A (a later parent's: not yet rewritten) -- come the same way, 16 lanes sharing an address.
    struct step_regs { int e, k; double x, uni, bias, a; };           // (k, x are read only where e >= 0: selected at the point of
                                                                       // use -- a select next to the load would wait for it)
    auto bounds = [&](const int p, int *s0, int *s1) {
        const int q = p + slot;
        *s0 = start[q < N ? q : N];                                    // start[N] == start[N + 1]: nothing past the last parent
        *s1 = start[q < N ? q + 1 : N + 1];
    };
    auto request = [&](const int p, const int code, const int s0, const int s1, step_regs &r, const bool in_loop = true) {
        const bool lone = code == 255;
        const int b0 = lone ? __builtin_amdgcn_readfirstlane(s0) + lane : s0 + sub;
        const int b1 = lone ? __builtin_amdgcn_readfirstlane(s1) : s1;
        const bool live = lone || slot < code;
        const int e = live && b0 < b1 ? b0 : -1;
        int pp = p + (!lone && slot < code ? slot : 0);
        pp = pp < N ? pp : N - 1;
        const unsigned at = e >= 0 ? (unsigned)e : 0u;
#if defined(ADJ_ABL) && (ADJ_ABL & 1)                                      /* timing experiments (tools/dbg/adjabl.sh), compile-time: */
        if (in_loop) { r.e = e; return; }                               /* a run-time switch would make the requests conditional  */
#endif
        r.e = e;
        r.k = ek[at];
        r.x = ex[at];
        r.uni = ug[(unsigned)pp];
        r.bias = bg[(unsigned)pp];
        r.a = Ac[(unsigned)pp];
    };
#ifdef NHP_STAMP
    unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // The chain walks the steps in order; what a step needs from global memory -- its entries {child slot, x} -- is data only,
    // and which entries those are follows from the grouping codes (LDS).  One global round trip is ~1500 cycles, a step's own
    // work ~850: the entries are therefore requested THREE steps ahead, into three register sets used round-robin (the loop
    // is unrolled by three so no set is ever copied: a register hand-over would wait for the load it carries).
    // Past the last parent the steps are empty groups (code 0: no live slot), so that the loop always runs whole rounds of
    // three: an exit between two steps would join the paths "requested just now" and "requested three steps ago", and the
    // compiler would again have to wait for everything.
    auto step_len = [&](const int q, const int cq) { return cq == 255 || cq == 0 ? 1 : cq; };
    auto code_at = [&](const int q) { return (int)grp[q < N ? q : N]; };
    // (the code read is wave-uniform, and the compiler would move it to a scalar register -- waiting for it -- where it is
    //  issued: it is kept a vector value until the step's work is done)
    auto settle_code = [&](int cv) { asm volatile("" : "+v"(cv)); return __builtin_amdgcn_readfirstlane(cv); };
#ifndef ADJ_DEPTH
#define ADJ_DEPTH 3
#endif
    constexpr int D = ADJ_DEPTH;
    int P[D], Cq[D];                                                   // heads and codes of the steps in flight (a ring; indices
    step_regs R[D];                                                    //  are compile-time after unrolling: all of it in registers)
    P[0] = 0; Cq[0] = code_at(0);
#pragma unroll
    for (int j = 1; j < D; ++j) { P[j] = P[j - 1] + step_len(P[j - 1], Cq[j - 1]); Cq[j] = code_at(P[j]); }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        int s0, s1;
        bounds(P[j], &s0, &s1);
        request(P[j], Cq[j], s0, s1, R[j], false);
    }
    const int NL = NHP_SKIP(a, 64) ? 0 : N;
#ifdef NHP_STAMP
    int n_steps = 0, n_general = 0;
    unsigned long long t_general = 0, t_group = 0;
#endif
    while (P[0] < NL) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            constexpr int dummy = 0; (void)dummy;
            const int last = (j + D - 1) % D;                           // the newest step in flight
            const int pn = P[last] + step_len(P[last], Cq[last]), cnv = code_at(pn);
            int s0n, s1n;
            bounds(pn, &s0n, &s1n);                                     // code and list bounds of step s + D: LDS reads under this step
            const int ck = R[j].e >= 0 ? R[j].k : -1;
            const double cx = R[j].e >= 0 ? R[j].x : 0.0;
#ifdef NHP_STAMP
            const unsigned long long tv0 = __builtin_amdgcn_s_memtime();
#endif
            if (Cq[j] == 255) visit_general(P[j], ck, cx, R[j].uni, R[j].bias, R[j].a);
            else visit_group(P[j], Cq[j], ck, cx, R[j].uni, R[j].bias, R[j].a);
#ifdef NHP_STAMP
            if (Cq[j] == 255) { ++n_general; t_general += __builtin_amdgcn_s_memtime() - tv0; }
            else if (Cq[j] != 0) t_group += __builtin_amdgcn_s_memtime() - tv0;
#endif
            const int cn = settle_code(cnv);
            request(pn, cn, s0n, s1n, R[j]);                            // step s + D into the set step s has just freed
            asm volatile("" ::: "memory");                              // keep the prefetch where it is issued
            P[j] = pn; Cq[j] = cn;
#ifdef NHP_STAMP
            ++n_steps;
#endif
        }
    }
    if (lane == 0 && col_links) col_links[c] = (double)links;
#ifdef NHP_STAMP
    if (lane == 0 && blockIdx.x < 1024) {
        g_adj_stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_begin;
        g_adj_stamps[4 * blockIdx.x + 1] = n_steps;
        g_adj_stamps[4 * blockIdx.x + 2] = n_general;
        g_adj_stamps[4 * blockIdx.x + 3] = t_general;
        if (blockIdx.x == 7) printf("col 7: %d steps, %d lone: visit cycles per step: lone %.0f, group %.0f\n", n_steps, n_general,
                                    (double)t_general / n_general, (double)t_group / (n_steps - n_general));
    }
#endif
}

// One sweep of A, enqueued on the ctx stream; the per-column link counts are left at *d_links_out [N] in the scratch.
// Link probabilities: rho_matrix (host, N*N) | d_rho_scalar (device scalar) | rho.
nhp_status nhp_adj_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m, const double *rho_matrix, double rho,
                           const double *d_rho_scalar, const double *u, uint64_t seed, uint64_t step, double **d_links_out)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (!m->has_A) { nhp_set_error(ctx, "resample_adjacency: the model has no adjacency matrix"); return NHP_EINVAL; }
    if (!rho_matrix && !d_rho_scalar && !(rho >= 0.0 && rho <= 1.0)) { nhp_set_error(ctx, "link probability must lie in [0, 1]"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, P = (size_t)(ds->pairs > 0 ? ds->pairs : 1);
    int max_children = 1;
    for (size_t c = 0; c < N; ++c) max_children = std::max(max_children, ds->h_boff[c + 1] - ds->h_boff[c]);
    const bool expo = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    const size_t lds_build = 4 * (2 * N + 2 + NHP_BLOCK + (size_t)max_children);
    const size_t lds_eval = (expo ? 16 : 24) * N + 8 * N + 8 * (size_t)max_children;
    const size_t lds_sweep = 20 * (size_t)max_children + 4 * (N + 2) + N + 8;
    if (lds_build > 160 * 1024 || lds_eval > 160 * 1024 || lds_sweep > 160 * 1024) {
        nhp_set_error(ctx, "resample_adjacency: a node with %d events (N = %d) exceeds the 160 KiB LDS column state", max_children, ds->N);
        return NHP_ENOTIMPL;
    }
    hipStream_t st = ctx->stream;
    nhp_cont_args a = nhp_make_args(ds, m);
    // a column shard sweeps its own columns: the columns of A are independent given the data (src/continuous.jl:444-470)
    const unsigned ncol = (unsigned)(ds->col_end - ds->col_begin);
    // ---- the data-only pair lists, built on first use and kept with the dataset
    if (!ds->d_adj_k) {
        nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
        if (hipMalloc((void **)&mds->d_adj_k, 4 * P) != hipSuccess || hipMalloc((void **)&mds->d_adj_p, 4 * P) != hipSuccess ||
            hipMalloc((void **)&mds->d_adj_dt, 8 * P) != hipSuccess || hipMalloc((void **)&mds->d_adj_start, 4 * N * (N + 1)) != hipSuccess ||
            hipMalloc((void **)&mds->d_adj_off, 8 * (N + 1)) != hipSuccess || hipMalloc((void **)&mds->d_adj_group, NN) != hipSuccess) {
            (void)hipFree(mds->d_adj_k); (void)hipFree(mds->d_adj_p); (void)hipFree(mds->d_adj_dt); (void)hipFree(mds->d_adj_start); (void)hipFree(mds->d_adj_off); (void)hipFree(mds->d_adj_group);
            mds->d_adj_group = nullptr; mds->d_adj_k = nullptr; mds->d_adj_p = nullptr; mds->d_adj_dt = nullptr; mds->d_adj_start = nullptr; mds->d_adj_off = nullptr;
            nhp_set_error(ctx, "resample_adjacency: out of device memory for %zu cached pairs", P);
            return NHP_ENOMEM;
        }
        NHP_HIP(ctx, hipMemcpyAsync(mds->d_adj_off, ds->h_pair_off.data(), 8 * (N + 1), hipMemcpyHostToDevice, st));
        if (lds_build > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_adj_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_build));
        hipLaunchKernelGGL(k_adj_build, dim3(ncol), dim3(NHP_BLOCK), lds_build, st, a, ds->d_adj_off, ds->group, mds->d_adj_k,
                           mds->d_adj_p, mds->d_adj_dt, mds->d_adj_start, mds->d_adj_group, max_children);
        NHP_HIP(ctx, hipGetLastError());
    }
    const size_t M1 = (size_t)(ds->M > 0 ? ds->M : 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t o_x = carve(8 * P), o_u = carve(8 * NN), o_rho = carve(8 * NN), o_links = carve(8 * N), o_lam = carve(8 * M1);
    const size_t o_uni = carve(8 * NN), o_bias = carve(8 * NN), o_sink = carve(8 * N);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, off));
    char *base = (char *)ctx->d_scratch;
    if (u) NHP_HIP(ctx, hipMemcpyAsync(base + o_u, u, 8 * NN, hipMemcpyHostToDevice, st));
    if (rho_matrix) NHP_HIP(ctx, hipMemcpyAsync(base + o_rho, rho_matrix, 8 * NN, hipMemcpyHostToDevice, st));
    const double *d_u = u ? (const double *)(base + o_u) : nullptr;
    const double *d_rho = rho_matrix ? (const double *)(base + o_rho) : nullptr;
    double *d_links = (double *)(base + o_links);
    if (ncol != (unsigned)N) NHP_HIP(ctx, hipMemsetAsync(d_links, 0, 8 * N, st));        // links of the columns this shard does not own
    const int64_t *d_off = ds->d_adj_off;
    const int32_t *d_k = ds->d_adj_k, *d_start = ds->d_adj_start;
    double *d_x = (double *)(base + o_x), *d_lam = (double *)(base + o_lam);
    double *d_uni = (double *)(base + o_uni), *d_bias = (double *)(base + o_bias);
    static const int eval_threads = getenv("NHP_ADJ_EVAL_THREADS") ? atoi(getenv("NHP_ADJ_EVAL_THREADS")) : 512;
#define NHP_AEVAL(imp, th, lqp)                                                                                        \
    do {                                                                                                               \
        if (lds_eval > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_adj_eval<imp, th>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval)); \
        hipLaunchKernelGGL((k_adj_eval<imp, th>), dim3(ncol), dim3(th), lds_eval, st, a, m->d_A, d_off, d_k, ds->d_adj_p, ds->d_adj_dt, lqp, d_x, \
                           max_children, d_lam, d_rho, rho, d_rho_scalar, d_u, seed, step, d_uni, d_bias);            \
    } while (0)
    if (expo) {
        if (eval_threads == 256) NHP_AEVAL(NHP_IMPULSE_EXPONENTIAL, 256, (const double2 *)nullptr);
        else if (eval_threads == 1024) NHP_AEVAL(NHP_IMPULSE_EXPONENTIAL, 1024, (const double2 *)nullptr);
        else NHP_AEVAL(NHP_IMPULSE_EXPONENTIAL, 512, (const double2 *)nullptr);
    } else {
        if (!ds->d_adj_lq) {                                         // first logit-normal sweep on this dataset: the data half of the pdf
            nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
            if (hipMalloc((void **)&mds->d_adj_lq, 16 * P) != hipSuccess) {
                mds->d_adj_lq = nullptr;
                nhp_set_error(ctx, "resample_adjacency: out of device memory for %zu cached pairs", P);
                return NHP_ENOMEM;
            }
            hipLaunchKernelGGL(k_adj_lq, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, (int64_t)ds->pairs, a.inv_dtmax, ds->d_adj_dt, mds->d_adj_lq);
        }
        if (eval_threads == 256) NHP_AEVAL(NHP_IMPULSE_LOGITNORMAL, 256, (const double2 *)ds->d_adj_lq);
        else if (eval_threads == 1024) NHP_AEVAL(NHP_IMPULSE_LOGITNORMAL, 1024, (const double2 *)ds->d_adj_lq);
        else NHP_AEVAL(NHP_IMPULSE_LOGITNORMAL, 512, (const double2 *)ds->d_adj_lq);
    }
#undef NHP_AEVAL
    NHP_HIP(ctx, hipGetLastError());
    if (lds_sweep > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_adj_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sweep));
    hipLaunchKernelGGL(k_adj_sweep, dim3(ncol), dim3(64), lds_sweep, st, a, m->d_A, d_off, d_k, d_x, d_start, ds->d_adj_group, d_lam,
                       d_uni, d_bias, max_children, d_links, (double *)(base + o_sink));
    NHP_HIP(ctx, hipGetLastError());
    *d_links_out = d_links;
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_resample_adjacency(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m,
                                                  const double *rho_matrix, double rho, const double *u,
                                                  uint64_t seed, uint64_t step, double *A_out, double *n_links)
{
    double *d_links = nullptr;
    NHP_TRY(nhp_adj_enqueue(ctx, ds, m, rho_matrix, rho, nullptr, u, seed, step, &d_links));
    const size_t N = (size_t)ds->N, NN = N * N;
    hipStream_t st = ctx->stream;
    std::vector<double> links(N);
    NHP_HIP(ctx, hipMemcpyAsync(links.data(), d_links, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    if (A_out) NHP_TRY(nhp_download(ctx, A_out, m->d_A, 8 * NN));
    if (n_links) {
        double s = 0.0;
        for (double v : links) s += v;
        *n_links = s;
    }
    return NHP_OK;
}


// ---- BernoulliNetworkModel on the device: resample!(network, A) (src/networks.jl:70-78) -----------------------------
// ρ ~ Beta(α + ΣA, β + N² - ΣA) as X / (X + Y), X ~ Gamma(α + ΣA), Y ~ Gamma(β + N² - ΣA), Philox-keyed (seed, step); the
// link counts of the columns are added in a fixed order (one workgroup: deterministic).  state = {ρ, Σρ, Σρ², ΣA}.
#include "nhp_rng.h"
__global__ __launch_bounds__(256) void k_links_total(const double *__restrict__ col_links, int N, double *__restrict__ state)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int c = threadIdx.x; c < N; c += 256) s += col_links[c];
    s = nhp_block_sum_n<4>(s, red);
    if (threadIdx.x == 0) state[3] = s;
}

__global__ void k_rho_draw(double *__restrict__ state, double alpha, double beta, double nn, uint64_t seed, uint64_t step)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double links = state[3];
    const double x = dev_gamma(alpha + links, 1.0, seed ^ 0x3F84D5B5B5470917ull, step, 0);
    const double y = dev_gamma(beta + nn - links, 1.0, seed ^ 0x3F84D5B5B5470917ull, step, 1);
    state[0] = x / (x + y);
}

static nhp_status ensure_rho(nhp_ctx *ctx, nhp_cont_model *m)
{
    if (m->d_rho) return NHP_OK;
    NHP_HIP(ctx, hipMalloc((void **)&m->d_rho, 4 * sizeof(double)));
    NHP_HIP(ctx, hipMemsetAsync(m->d_rho, 0, 4 * sizeof(double), ctx->stream));
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_set_rho(nhp_ctx *ctx, nhp_cont_model *m, double rho)
{
    if (!ctx || !m) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!(rho >= 0.0 && rho <= 1.0)) { nhp_set_error(ctx, "link probability must lie in [0, 1]"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(ensure_rho(ctx, m));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_rho, &rho, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));          // `rho` is a stack value
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_get_rho(nhp_ctx *ctx, const nhp_cont_model *m, double *out)
{
    if (!ctx || !m || !out) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!m->d_rho) { nhp_set_error(ctx, "get_rho: the model has no device-side link probability (nhp_cont_model_set_rho)"); return NHP_EINVAL; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    return nhp_download(ctx, out, m->d_rho, 3 * sizeof(double));
}

extern "C" nhp_status nhp_cont_network_step(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, nhp_cont_model *m,
                                            double alpha, double beta, uint64_t seed, uint64_t step)
{
    if (!ctx || !m) return NHP_EINVAL;
    if (!m->d_rho) { nhp_set_error(ctx, "network_step: set the link probability first (nhp_cont_model_set_rho)"); return NHP_EINVAL; }
    const bool fixed = alpha == 0.0 && beta == 0.0;                                         // DenseNetworkModel: ρ stays
    if (!fixed && !(alpha > 0.0 && beta > 0.0)) { nhp_set_error(ctx, "network_step: the Beta prior needs alpha, beta > 0"); return NHP_EDOMAIN; }
    if (!comm && ds && nhp_is_column_shard(ds)) { nhp_set_error(ctx, "network_step: a column shard needs the communicator of its ranks"); return NHP_EINVAL; }
    double *d_links = nullptr;
    NHP_TRY(nhp_adj_enqueue(ctx, ds, m, nullptr, 0.5, m->d_rho, nullptr, seed, step, &d_links));
    hipLaunchKernelGGL(k_links_total, dim3(1), dim3(256), 0, ctx->stream, d_links, ds->N, m->d_rho);
    NHP_HIP(ctx, hipGetLastError());
    if (comm) NHP_TRY(nhp_comm_allreduce_dev(ctx, comm, m->d_rho + 3, 1));      // the shards' link counts (a one-rank clique runs the same call)
    const double nn = (double)ds->N * (double)ds->N;
    if (!fixed) {
        hipLaunchKernelGGL(k_rho_draw, dim3(1), dim3(64), 0, ctx->stream, m->d_rho, alpha, beta, nn, seed, step);
        NHP_HIP(ctx, hipGetLastError());
    }
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_network_sweep(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m, uint64_t seed, uint64_t step,
                                             double *n_links)
{
    if (!ctx || !m || !n_links) return NHP_EINVAL;
    if (!m->d_rho) { nhp_set_error(ctx, "network_sweep: set the link probability first (nhp_cont_model_set_rho)"); return NHP_EINVAL; }
    double *d_links = nullptr;
    NHP_TRY(nhp_adj_enqueue(ctx, ds, m, nullptr, 0.5, m->d_rho, nullptr, seed, step, &d_links));
    hipLaunchKernelGGL(k_links_total, dim3(1), dim3(256), 0, ctx->stream, d_links, ds->N, m->d_rho);
    NHP_HIP(ctx, hipGetLastError());
    return nhp_download(ctx, n_links, m->d_rho + 3, sizeof(double));
}

extern "C" nhp_status nhp_cont_network_rho(nhp_ctx *ctx, nhp_cont_model *m, double alpha, double beta, double n_links, double n_entries,
                                           uint64_t seed, uint64_t step)
{
    if (!ctx || !m) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!m->d_rho) { nhp_set_error(ctx, "network_rho: set the link probability first (nhp_cont_model_set_rho)"); return NHP_EINVAL; }
    if (alpha == 0.0 && beta == 0.0) return NHP_OK;
    if (!(alpha > 0.0 && beta > 0.0) || !(n_links >= 0.0 && n_links <= n_entries)) { nhp_set_error(ctx, "network_rho: bad Beta parameters"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_rho + 3, &n_links, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rho_draw, dim3(1), dim3(64), 0, ctx->stream, m->d_rho, alpha, beta, n_entries, seed, step);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));                       // n_links is a stack value
    return NHP_OK;
}

// ---- chain driver: the loop body of mcmc! (src/inference.jl:55-62), device-resident end to end ------------------------
extern "C" nhp_status nhp_cont_mcmc_run(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, nhp_cont_model *m,
                                        const nhp_gibbs_priors *pr, double net_alpha, double net_beta, uint64_t seed,
                                        uint64_t step0, int64_t n_steps, int64_t burn)
{
    if (!ctx || !m || !pr || n_steps < 0) return NHP_EINVAL;
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (m->has_A && !m->d_rho) { nhp_set_error(ctx, "mcmc_run: set the network's link probability first (nhp_cont_model_set_rho)"); return NHP_EINVAL; }
    for (int64_t k = 0; k < n_steps; ++k) {
        const uint64_t step = step0 + (uint64_t)k;
        NHP_TRY(nhp_cont_gibbs_step(ctx, ds, m, pr, seed, step));           // (reports a sampler error one sweep late)
        if (m->has_A) NHP_TRY(nhp_cont_network_step(ctx, comm, ds, m, net_alpha, net_beta, seed, step));
        if (burn >= 0 && (int64_t)step >= burn) NHP_TRY(nhp_cont_model_moments_accumulate(ctx, m));
    }
    return nhp_ctx_synchronize(ctx);
}
