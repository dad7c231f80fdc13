// Gibbs sweep of the adjacency matrix (reference: resample_adjacency_matrix! / resample_column!
// src/continuous.jl:444-487, integrated_intensity :489-498, sum_log_intensity :500-519,
// logsumexp src/utils/helpers.jl:13-16, link_probability src/networks.jl:65-68).
//
// The reference evaluates, for each of the N² entries, two full passes over every event of the child
// node (2·N² event scans per sweep).  Columns are independent; inside column c the entries are
// sampled one parent node after the other, each conditional on the current state of the column.
//
// Here workgroup c owns column c.  Phase 1 evaluates every parent-child pair of the column ONCE,
// x = W[p,c]·ħ(Δt), and counting-sorts the pairs by parent node p into (child slot, x) lists, while
// λ_k = λ0 + Σ A[p,c]·x is accumulated per child in LDS.  Phase 2 walks p = 1..N: only the children
// that have a parent on node p are touched --
//     ll1 - ll0 = -W[p,c]·cnt[p] + Σ_k [log(λ_k⁰ + x_kp) - log λ_k⁰] + log ρ - log(1-ρ),
// (the baseline integral and the first event's dropped term, SURVEY D10, cancel) -- the entry is
// drawn with the reference's Bernoulli rule u <= exp(ll1 - logsumexp(ll0, ll1)), and the affected
// λ_k are updated.  Work per sweep: Σ pairs instead of 2·N²·M.
#include <algorithm>

#include "nhp_internal.h"
#include "nhp_math.h"

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

template <int IMP>
__global__ __launch_bounds__(NHP_BLOCK) void k_adjacency(nhp_cont_args a, double *__restrict__ A,
                                                         const int64_t *__restrict__ pair_off,
                                                         int32_t *__restrict__ ent_k, double *__restrict__ ent_x,
                                                         const double *__restrict__ rho_mat, double rho_scalar,
                                                         const double *__restrict__ u, uint64_t seed, uint64_t step,
                                                         int max_children, double *__restrict__ col_links)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double2 *col = reinterpret_cast<double2 *>(smem + 64);                 // [N] exp {θ, W}; logit {μ, √τ}
    double *colw = reinterpret_cast<double *>(col + N);                    // [N] logit: W
    double *lam = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : N);         // [max_children] current λ_k
    double *dx = lam + max_children;                                       // [max_children] Σ x_kp of the current p
    int *marker = reinterpret_cast<int *>(dx + max_children);              // [max_children]
    int *start = marker + max_children;                                    // [N + 1] pair-list offsets by p
    int *cursor = start + N + 1;                                           // [N]
    int *scan_tmp = cursor + N;                                            // [NHP_BLOCK]

    const int kb = a.boff[c], ke = a.boff[c + 1], nchild = ke - kb;
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            col[p] = make_double2(a.p1[k], a.W[k]);
        } else {
            col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
            colw[p] = a.W[k];
        }
        cursor[p] = 0;
    }
    for (int k = tid; k < nchild; k += NHP_BLOCK) {
        lam[k] = adj_baseline(a, c, a.child[kb + k].t);
        dx[k] = 0.0;
        marker[k] = 0;
    }
    __syncthreads();

    // ---- phase 1a: pairs per parent node
    for (int k = wave; k < nchild; k += NHP_WAVES) {
        const nhp_child ch = a.child[kb + k];
        for (int j = ch.first + lane; j < ch.idx; j += 64) atomicAdd(&cursor[a.nodes[j]], 1);
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

    // ---- phase 1b: evaluate every pair once, scatter by parent node, accumulate λ_k
    const int64_t base = pair_off[c];
    for (int k = wave; k < nchild; k += NHP_WAVES) {
        const nhp_child ch = a.child[kb + k];
        for (int j = ch.first + lane; j < ch.idx; j += 64) {
            const int p = a.nodes[j];
            const double dt = ch.t - a.times[j];
            const double2 q = col[p];
            double x;
            if (IMP == NHP_IMPULSE_EXPONENTIAL) x = q.y * nhp_pdf_exponential(q.x, dt);
            else x = colw[p] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
            const int pos = start[p] + atomicAdd(&cursor[p], 1);
            ent_k[base + pos] = k;
            ent_x[base + pos] = x;
            const double av = A[(size_t)p + (size_t)c * N];
            if (av != 0.0) atomicAdd(&lam[k], av * x);
        }
    }
    __syncthreads();

    // ---- phase 2: sequential Gibbs over parent nodes.  A parent node touches only a handful of
    // children, so ONE wave walks the column: no block barriers on the 1..N critical path, only
    // wave-local LDS ordering (LDS operations of a wave complete in issue order; the fence makes the
    // atomics of all lanes visible before any lane reads them back).  The per-entry constants --
    // uniform draw, prior log-odds, -W·cnt -- are precomputed by the whole block into LDS.
    double *uni = reinterpret_cast<double *>(smem + (((reinterpret_cast<unsigned char *>(scan_tmp + NHP_BLOCK) - smem) + 7) & ~(size_t)7));   // [N] logit of the draw u[p,c]
    double *bias = uni + N;                                                // [N] log ρ - log(1-ρ) - W·cnt[p]
    double *acol = bias + N;                                               // [N] current A[·,c]
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t kpc = (size_t)p + (size_t)c * N;
        const double rho = rho_mat ? rho_mat[kpc] : rho_scalar;
        const double w = IMP == NHP_IMPULSE_EXPONENTIAL ? col[p].y : colw[p];
        // the Bernoulli rule u <= exp(ll1 - logsumexp(ll0, ll1)) = 1/(1 + e^{-d}) is logit(u) <= d:
        // the logit is taken here, in parallel, so the serial chain below carries no exp and no division
        const double uu = u ? u[kpc] : nhp_philox_uniform(seed ^ 0xBE5466CF34E90C6Cull, step, kpc);
        uni[p] = nhp_log(uu / (1.0 - uu));
        bias[p] = -(w * a.cnt[p]) + nhp_log(rho) - nhp_log(1.0 - rho);
        acol[p] = A[kpc];
    }
    __syncthreads();
    if (wave != 0) return;
    double links = 0.0;
    // entries of parent p: lane l owns entry eb + l (+64, +128, ... for the rare long lists).  The first
    // chunk of parent p+1 is fetched while p is processed, so no global-load latency sits on the chain.
    int nk = -1;
    double nx = 0.0;
    if (N > 0 && start[0] + lane < start[1]) { nk = ent_k[base + start[0] + lane]; nx = ent_x[base + start[0] + lane]; }
    for (int p = 0; p < (NHP_SKIP(a, 64) ? 0 : N); ++p) {
        const int eb = start[p], ee = start[p + 1];
        const size_t kpc = (size_t)p + (size_t)c * N;
        const double aold = acol[p];
        const int ck = nk;                       // this lane's first entry of p (or -1)
        const double cx = nx;
        nk = -1;
        if (p + 1 < N && ee + lane < start[p + 2]) { nk = ent_k[base + ee + lane]; nx = ent_x[base + ee + lane]; }
        if (ck >= 0) atomicAdd(&dx[ck], cx);
        for (int e = eb + 64 + lane; e < ee; e += 64) atomicAdd(&dx[ent_k[base + e]], ent_x[base + e]);
        NHP_LDS_SYNC();
        double delta = 0.0;
        auto own = [&](int k) {
            if (atomicExch(&marker[k], p + 1) != p + 1) {           // first entry of child k for this p owns it
                const double d = dx[k];
                const double l0 = lam[k] - aold * d;
                delta += nhp_log(l0 + d) - nhp_log(l0);
            }
        };
        if (ck >= 0) own(ck);
        for (int e = eb + 64 + lane; e < ee; e += 64) own(ent_k[base + e]);
        delta = nhp_wave_sum(delta);
        const double d = bias[p] + delta;
        const double anew = uni[p] <= d ? 1.0 : 0.0;             // ll1 - ll0 = d
        if (lane == 0) A[kpc] = anew;
        links += anew;
        auto settle = [&](int k) {
            // the first taker carries the child's total
            const double dd = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long *>(&dx[k]), 0ull));
            if (dd != 0.0 && anew != aold) lam[k] += (anew - aold) * dd;
        };
        if (ck >= 0) settle(ck);
        for (int e = eb + 64 + lane; e < ee; e += 64) settle(ent_k[base + e]);
        NHP_LDS_SYNC();
    }
    if (tid == 0 && col_links) col_links[c] = links;
}

extern "C" nhp_status nhp_cont_resample_adjacency(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m,
                                                  const double *rho_matrix, double rho, const double *u,
                                                  uint64_t seed, uint64_t step, double *A_out, double *n_links)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (!m->has_A) { nhp_set_error(ctx, "resample_adjacency: the model has no adjacency matrix"); return NHP_EINVAL; }
    if (!rho_matrix && !(rho >= 0.0 && rho <= 1.0)) { nhp_set_error(ctx, "link probability must lie in [0, 1]"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, P = (size_t)(ds->pairs > 0 ? ds->pairs : 1);
    int max_children = 1;
    for (size_t c = 0; c < N; ++c) max_children = std::max(max_children, ds->h_boff[c + 1] - ds->h_boff[c]);
    const size_t per = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? 16 : 24;
    const size_t lds = 64 + per * N + 20 * (size_t)max_children + 4 * (2 * N + 2 + NHP_BLOCK) + 24 * N + 16;
    if (lds > 160 * 1024) {
        nhp_set_error(ctx, "resample_adjacency: a node with %d events (N = %d) exceeds the 160 KiB LDS column state", max_children, ds->N);
        return NHP_ENOTIMPL;
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t o_k = carve(4 * P), o_x = carve(8 * P), o_off = carve(8 * (N + 1)), o_u = carve(8 * NN), o_rho = carve(8 * NN), o_links = carve(8 * N);
    if (off > ((size_t)48 << 30)) { nhp_set_error(ctx, "resample_adjacency: %zu pairs need too much scratch", P); return NHP_ENOMEM; }
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, off));
    char *base = (char *)ctx->d_scratch;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(base + o_off, ds->h_pair_off.data(), 8 * (N + 1), hipMemcpyHostToDevice, st));
    if (u) NHP_HIP(ctx, hipMemcpyAsync(base + o_u, u, 8 * NN, hipMemcpyHostToDevice, st));
    if (rho_matrix) NHP_HIP(ctx, hipMemcpyAsync(base + o_rho, rho_matrix, 8 * NN, hipMemcpyHostToDevice, st));
    nhp_cont_args a = nhp_make_args(ds, m);
    const double *d_u = u ? (const double *)(base + o_u) : nullptr;
    const double *d_rho = rho_matrix ? (const double *)(base + o_rho) : nullptr;
    double *d_links = (double *)(base + o_links);
    if (m->impulse_kind == NHP_IMPULSE_EXPONENTIAL) {
        if (lds > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_adjacency<NHP_IMPULSE_EXPONENTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_adjacency<NHP_IMPULSE_EXPONENTIAL>), dim3((unsigned)N), dim3(NHP_BLOCK), lds, st, a, m->d_A,
                           (const int64_t *)(base + o_off), (int32_t *)(base + o_k), (double *)(base + o_x), d_rho, rho, d_u, seed, step,
                           max_children, d_links);
    } else {
        if (lds > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_adjacency<NHP_IMPULSE_LOGITNORMAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_adjacency<NHP_IMPULSE_LOGITNORMAL>), dim3((unsigned)N), dim3(NHP_BLOCK), lds, st, a, m->d_A,
                           (const int64_t *)(base + o_off), (int32_t *)(base + o_k), (double *)(base + o_x), d_rho, rho, d_u, seed, step,
                           max_children, d_links);
    }
    NHP_HIP(ctx, hipGetLastError());
    std::vector<double> links(N);
    NHP_HIP(ctx, hipMemcpyAsync(links.data(), d_links, 8 * N, hipMemcpyDeviceToHost, st));
    if (A_out) NHP_HIP(ctx, hipMemcpyAsync(A_out, m->d_A, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    if (n_links) {
        double s = 0.0;
        for (double v : links) s += v;
        *n_links = s;
    }
    return NHP_OK;
}
