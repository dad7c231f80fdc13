// Parent-assignment sampler and Gibbs sufficient statistics (reference:
// resample_parents / resample_parent src/parents.jl:1-46; node_counts / parent_counts
// :61-79; baseline node_counts src/baselines.jl:87-96; duration_mean src/impulses.jl:84-96;
// log_duration_sum / log_duration_variation :228-252).
//
// Sampler.  One categorical draw per event over the weights
//     [A·W·ħ(t_i - t_{i-1}), A·W·ħ(t_i - t_{i-2}), ..., λ0_{c_i}(t_i)]
// (most recent parent first, baseline last), normalised by their sum, with the inverse-CDF
// rule of Distributions.jl: smallest k with cumsum_k > u, capped at the last entry.
// BASELINE.json asks for indices that are bit-exact for a fixed uniform stream, so a child is
// owned by ONE lane that (1) sums the weights in exactly the order Julia's `sum` uses on the
// reference's Vector{Any} -- sequential up to 1024 entries, midpoint-split pairwise above --
// and (2) rescans them in order; every weight is evaluated with the fixed operation sequence
// of nhp_math.h.  The look-back window makes this embarrassingly parallel over events
// (children are node-bucketed so column c of the parameter tables sits in LDS).
// Uniforms: a caller-supplied array, or Philox4x32-10 keyed (seed, step, event index).
//
// Statistics.  The sampler also writes, in bucket order, each child's parent node and the
// child-parent delay.  A second kernel gives workgroup c the children of node c and lets
// lane p scan them for parent node p: no atomics, every (p,c) cell accumulates in child time
// order -- the order the reference's serial loops use -- so ΣΔt is reproducible bit for bit.
#include <algorithm>
#include <type_traits>
#include "nhp_internal.h"
#include "nhp_math.h"
#include "nhp_rng.h"

struct samp_col {                  // staged column c of the parameter tables
    const double2 *col;            // exp: {rate, a*w};  logit-normal: {mu, sqrt(tau)}
    const double *colw;            // logit-normal: a*w
};

template <int IMP>
__device__ __forceinline__ double samp_weight(const nhp_cont_args &a, const samp_col &sc, double t, int j)
{
#pragma clang fp contract(off)
    const nhp_event e = a.ev[j];                  // one 16-byte load: (t_j, n_j)
    const double dt = t - e.t;
    const int p = e.node;
    const double2 q = sc.col[p];
    if (IMP == NHP_IMPULSE_EXPONENTIAL) return q.y * nhp_pdf_exponential(q.x, dt);
    return sc.colw[p] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
}

// Weight k of a child through the logit-normal pair cache (lq, nd already point at the child's first pair): the logarithm and
// the division of the pdf were taken when the cache was built, with the same operations -- the same bits.
__device__ __forceinline__ double samp_weight_cached(const samp_col &sc, const double2 d, const int p)
{
#pragma clang fp contract(off)
    const double2 q = sc.col[p];
    return sc.colw[p] * nhp_pdf_logitnormal_cached(q.x, q.y, d);
}

__global__ __launch_bounds__(256) void k_plq_build(nhp_cont_args a, double2 *__restrict__ plq, uint16_t *__restrict__ pnode)
{
#pragma clang fp contract(off)
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.M) return;
    const nhp_child ch = a.child_w[k];
    const uint32_t o = a.poff[k];
    for (int r = 0; r < ch.idx - ch.first; ++r) {
        const nhp_event e = a.ev[ch.idx - 1 - r];
        plq[(size_t)o + r] = nhp_logitnormal_data(a.inv_dtmax, ch.t - e.t);
        pnode[(size_t)o + r] = (uint16_t)e.node;
    }
}

__device__ __forceinline__ double samp_baseline(const nhp_cont_args &a, int c, double t)
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

// weight k of child i: k < n-1 -> parent i-1-k, k == n-1 -> baseline
#define SAMP_W(k) ((k) < n - 1 ? (IMP != NHP_IMPULSE_EXPONENTIAL && lq ? samp_weight_cached(sc, lq[k], nd[k]) : samp_weight<IMP>(a, sc, t, i - 1 - (k))) : base)

// Sequential left fold over [lo, hi] -- Julia's mapreduce_impl leaf (and the n < 16 path).
#define SAMP_CACHE 16      // weights per child kept in LDS between the sum pass and the scan pass (8: 133 us vs 122 us at mean window 8)
#define SAMP_CLD 17        // row stride (doubles): odd, so lanes reading the same k hit distinct banks

template <int IMP>
__device__ __forceinline__ double samp_weight_of(const nhp_cont_args &a, const samp_col &sc, double t, const nhp_event &e)
{
#pragma clang fp contract(off)
    const double dt = t - e.t;
    const double2 q = sc.col[e.node];
    if (IMP == NHP_IMPULSE_EXPONENTIAL) return q.y * nhp_pdf_exponential(q.x, dt);
    return sc.colw[e.node] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
}

// Four weights at a time: their parent records are requested together and the four pdf evaluations are independent
// instruction streams; only the additions stay in sequence (the order is the contract).  Slots past `hi` or at the
// baseline position evaluate a clamped record whose value is discarded.
#define SAMP_FU 4
template <int IMP>
__device__ __forceinline__ double samp_fold(const nhp_cont_args &a, const samp_col &sc, double t, int i,
                                            int n, double base, int lo, int hi, double *wcache,
                                            const double2 *lq, const uint16_t *nd)
{
#pragma clang fp contract(off)
    double v = 0.0;
    for (int k0 = lo; k0 <= hi; k0 += SAMP_FU) {
        double w[SAMP_FU];
        if (IMP != NHP_IMPULSE_EXPONENTIAL && lq) {                  // (wave-uniform) the cached data half of the pdf
            double2 d[SAMP_FU];
            int p[SAMP_FU];
#pragma unroll
            for (int u = 0; u < SAMP_FU; ++u) { const int k = k0 + u < n - 1 ? k0 + u : 0; d[u] = lq[k]; p[u] = nd[k]; }
#pragma unroll
            for (int u = 0; u < SAMP_FU; ++u) w[u] = samp_weight_cached(sc, d[u], p[u]);
        } else {
            nhp_event e[SAMP_FU];
#pragma unroll
            for (int u = 0; u < SAMP_FU; ++u) e[u] = a.ev[k0 + u < n - 1 ? i - 1 - (k0 + u) : i - 1];
#pragma unroll
            for (int u = 0; u < SAMP_FU; ++u) w[u] = samp_weight_of<IMP>(a, sc, t, e[u]);
        }
#pragma unroll
        for (int u = 0; u < SAMP_FU; ++u) {
            const int k = k0 + u;
            if (k <= hi) {
                const double wk = k < n - 1 ? w[u] : base;
                if (k < SAMP_CACHE) wcache[k] = wk;
                v = k == lo ? wk : v + wk;
            }
        }
    }
    return v;
}

// Julia Base `sum` over n boxed elements: sequential for n <= 1024, otherwise split at
// lo + (hi-lo)>>1 recursively (reduce.jl, pairwise_blocksize = 1024).  Iterative post-order.
template <int IMP>
__device__ double samp_sum(const nhp_cont_args &a, const samp_col &sc, double t, int i, int n, double base, double *wcache,
                           const double2 *lq, const uint16_t *nd)
{
#pragma clang fp contract(off)
    if (n <= 1024) return samp_fold<IMP>(a, sc, t, i, n, base, 0, n - 1, wcache, lq, nd);
    int s_lo[24], s_hi[24], s_state[24];
    double s_left[24];
    int sp = 0;
    double ret = 0.0;
    s_lo[0] = 0; s_hi[0] = n - 1; s_state[0] = 0; sp = 1;
    while (sp > 0) {
        const int f = sp - 1;
        const int lo = s_lo[f], hi = s_hi[f];
        if (s_state[f] == 0) {
            if (hi - lo < 1024) {
                ret = samp_fold<IMP>(a, sc, t, i, n, base, lo, hi, wcache, lq, nd);
                --sp;
            } else {
                s_state[f] = 1;
                s_lo[sp] = lo; s_hi[sp] = lo + ((hi - lo) >> 1); s_state[sp] = 0; ++sp;
            }
        } else if (s_state[f] == 1) {
            s_left[f] = ret;
            s_state[f] = 2;
            s_lo[sp] = lo + ((hi - lo) >> 1) + 1; s_hi[sp] = hi; s_state[sp] = 0; ++sp;
        } else {
            ret = s_left[f] + ret;
            --sp;
        }
    }
    return ret;
}

// Logit-normal impulses on a short-window dataset: the data half of every pair's pdf, made once (k_plq_build) and shared by
// the parent sampler and the log-likelihood kernel.  NHP_EINVAL when the dataset has no pair offsets, the cache is switched
// off (NHP_PLQ=0) or there is no room for it; the callers then evaluate the whole pdf per pair.
nhp_status nhp_ensure_pair_cache(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_args *a)
{
    if (!ds->d_poff || (getenv("NHP_PLQ") && atoi(getenv("NHP_PLQ")) == 0)) { a->plq = nullptr; a->pnode = nullptr; return NHP_EINVAL; }
    if (!ds->d_plq) {
        nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
        const size_t P = (size_t)std::max<int64_t>(ds->pairs, 1);
        if (hipMalloc((void **)&mds->d_plq, 16 * P) != hipSuccess || hipMalloc((void **)&mds->d_pnode, 2 * P) != hipSuccess) {
            (void)hipFree(mds->d_plq); mds->d_plq = nullptr; mds->d_pnode = nullptr;
            (void)hipGetLastError();
            a->plq = nullptr; a->pnode = nullptr;
            return NHP_EINVAL;
        }
        hipLaunchKernelGGL(k_plq_build, dim3((unsigned)((ds->M + 255) / 256)), dim3(256), 0, ctx->stream, *a, mds->d_plq, mds->d_pnode);
        NHP_HIP(ctx, hipGetLastError());
    }
    a->plq = ds->d_plq; a->pnode = ds->d_pnode;
    return NHP_OK;
}

template <int IMP>
__global__ __launch_bounds__(NHP_BLOCK) void k_sampler(nhp_cont_args a, const double *__restrict__ u,
                                                       uint64_t seed, uint64_t step,
                                                       int64_t *__restrict__ parents, int64_t *__restrict__ pnodes,
                                                       int32_t *__restrict__ pn_b, double *__restrict__ dt_b,
                                                       int *__restrict__ err)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *col = reinterpret_cast<double2 *>(smem);
    double *colw = reinterpret_cast<double *>(col + a.N);
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x;
    // the first SAMP_CACHE weights of a child survive from the sum pass to the scan pass in LDS: at
    // short windows the scan then costs no second round of exp / log evaluations (same values, so the
    // indices stay bit-exact)
    double *wcache = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : a.N) + (size_t)tid * SAMP_CLD;
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        double w = a.W[k];
        if (a.A) w = a.A[k] * w;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            const double scale = 1.0 / a.p1[k];            // Exponential(1/θ) ...
            col[p] = make_double2(1.0 / scale, w);         // ... and its rate inv(scale)
        } else {
            col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
            colw[p] = w;
        }
    }
    __syncthreads();
    samp_col sc{col, colw};

    // children in the window-sorted order of the log-likelihood kernels: the lanes of a wave then walk windows of
    // (nearly) equal length instead of waiting for the longest of 64; results go to the child's bucket position
    for (int kw = it.kbeg + tid; kw < it.kend; kw += NHP_BLOCK) {
        const nhp_child ch = a.child_w[kw];
        const int k = a.wpos[kw];
        const int i = ch.idx;
        const double t = ch.t;
        int parent = -1;
        if (i > 0) {                                       // index == 1 -> (0, 0): src/parents.jl:26-28
            const int n = i - ch.first + 1;
            const double base = samp_baseline(a, c, t);
            const double2 *lq = (IMP != NHP_IMPULSE_EXPONENTIAL && a.plq) ? a.plq + a.poff[kw] : nullptr;
            const uint16_t *nd = lq ? a.pnode + a.poff[kw] : nullptr;
            const double s = samp_sum<IMP>(a, sc, t, i, n, base, wcache, lq, nd);
            if (!(s > 0.0) || !(s < __builtin_inf())) *err = 1;
            const double draw = u ? u[i] : nhp_philox_uniform(seed, step, (uint64_t)i);
            int kk = 0;
            double cp = wcache[0] / s;
            while (cp <= draw && kk < n - 1) {
                ++kk;
                cp = cp + (kk < SAMP_CACHE ? wcache[kk] : SAMP_W(kk)) / s;
            }
            if (kk < n - 1) parent = i - 1 - kk;
        }
        const int pnode = parent >= 0 ? a.nodes[parent] : -1;
        if (parents) parents[i] = (int64_t)parent + 1;     // 1-based event index, 0 = baseline
        if (pnodes) pnodes[i] = (int64_t)pnode + 1;
        pn_b[k] = pnode;
        dt_b[k] = parent >= 0 ? t - a.times[parent] : 0.0;
    }
}

// ---- 8 lanes per child ------------------------------------------------------------------------------
// k_sampler gives a child to one lane; its window is then walked with 64 different cache lines per load
// instruction and one dependent load per parent.  Here 8 lanes share a child and a window is consumed 8
// records at a time -- one contiguous 128-byte run per child and instruction, the access pattern of the
// windowed log-likelihood kernel -- while the order-sensitive arithmetic stays EXACTLY the single-lane
// sequence: each lane evaluates one weight, then the partial sums ((w0 + w1) + w2) + ... are formed by a
// chain of eight dependent steps, step t executed by lane t on the value lane t-1 produced (DPP row_shr:1),
// and the cumulative probabilities cp_k = cp_{k-1} + w_k/s the same way (one division per lane, in
// parallel).  Bit-identical indices; used when no window exceeds Julia's 1024-element sequential-sum limit.
#define S8_CACHE 4       // chunks of 8 weights kept in registers between the sum pass and the scan pass

__device__ __forceinline__ double s8_shr1(double v)        // lane i <- lane i-1 (within a row of 16)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xF, 0xF, false));
}

// sequential prefix over the 8 lanes of a group: returns p_k = (...((carry ⊕ x_0) + x_1)...) + x_k in lane k,
// where `carry ⊕ x_0` is x_0 itself for the first chunk (the reference starts from the first element, not 0)
__device__ __forceinline__ double s8_chain(double x, double carry, bool first, int gl)
{
#pragma clang fp contract(off)
    double v = first ? x : carry + x;                       // lane 0's value; other lanes overwrite below
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        const double prev = s8_shr1(v);
        if (gl == t) v = prev + x;
    }
    return v;
}

template <int IMP>
__global__ __launch_bounds__(NHP_BLOCK) void k_sampler8(nhp_cont_args a, const double *__restrict__ u,
                                                        uint64_t seed, uint64_t step,
                                                        int64_t *__restrict__ parents, int64_t *__restrict__ pnodes,
                                                        int32_t *__restrict__ pn_b, double *__restrict__ dt_b,
                                                        int *__restrict__ err)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *col = reinterpret_cast<double2 *>(smem);
    double *colw = reinterpret_cast<double *>(col + a.N);
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x, gl = tid & 7, gid = tid >> 3;
    const int glane7 = (tid & 63 & ~7) + 7;                  // last lane of this group inside the wave
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        double w = a.W[k];
        if (a.A) w = a.A[k] * w;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            const double scale = 1.0 / a.p1[k];            // Exponential(1/θ) ...
            col[p] = make_double2(1.0 / scale, w);         // ... and its rate inv(scale)
        } else {
            col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
            colw[p] = w;
        }
    }
    __syncthreads();
    samp_col sc{col, colw};

    const int nchild = it.kend - it.kbeg;
    for (int k0 = 0; k0 < nchild; k0 += NHP_BLOCK / 8) {    // block-uniform trip count; idle groups are masked
        const int kq = k0 + gid;
        const bool live = kq < nchild;
        const nhp_child ch = a.child_w[it.kbeg + (live ? kq : 0)];      // window-sorted order: equal chunk counts per wave
        const int i = ch.idx;
        const double t = ch.t;
        const int n = (live && i > 0) ? i - ch.first + 1 : 0;           // index == 1 -> (0, 0): src/parents.jl:26-28
        const double base = n > 0 ? samp_baseline(a, c, t) : 0.0;
        // chunks needed by any group of this wave (groups walk in lock step; a finished group adds zeros)
        int nq = (n + 7) >> 3;
        int nq_max = nq;
#pragma unroll
        for (int o = 32; o >= 8; o >>= 1) nq_max = max(nq_max, __shfl_xor(nq_max, o));
        // weight k of the child: k < n-1 -> parent i-1-k, k == n-1 -> baseline, beyond -> 0
        auto weight = [&](int k) -> double {
            if (k < n - 1) return samp_weight<IMP>(a, sc, t, i - 1 - k);
            return k == n - 1 ? base : 0.0;
        };
        // ---- pass 1: s = w_0 + w_1 + ... left to right (the first S8_CACHE chunks stay in registers: static
        // indices, so they really are registers)
        double wreg[S8_CACHE];
        double carry = 0.0;
#pragma unroll
        for (int q = 0; q < S8_CACHE; ++q) {
            wreg[q] = 0.0;
            if (q < nq_max) {
                wreg[q] = weight(8 * q + gl);
                const double v = s8_chain(wreg[q], carry, q == 0, gl);
                carry = __shfl(v, glane7);
            }
        }
        for (int q = S8_CACHE; q < nq_max; ++q) {
            const double v = s8_chain(weight(8 * q + gl), carry, false, gl);
            carry = __shfl(v, glane7);
        }
        const double s = carry;
        if (n > 0 && (!(s > 0.0) || !(s < __builtin_inf()))) *err = 1;
        // ---- pass 2: smallest kk with cp_kk > draw, capped at n-1 (the baseline)
        const double draw = n > 0 ? (u ? u[i] : nhp_philox_uniform(seed, step, (uint64_t)i)) : 0.0;
        int kk = n > 0 ? n - 1 : 0;
        bool found = n == 0;
        carry = 0.0;
        auto scan = [&](const int q, const double w) {
            const double cp = s8_chain(w / s, carry, q == 0, gl);
            carry = __shfl(cp, glane7);
            const int kidx = 8 * q + gl;
            // lanes of the group whose cumulative probability already exceeds the draw (only real weights count)
            const unsigned long long hits = __ballot(!found && kidx < n && cp > draw);
            const unsigned int mine = (unsigned int)(hits >> ((tid & 63) & ~7)) & 0xFFu;
            if (!found && mine) { kk = 8 * q + (__ffs(mine) - 1); found = true; }
        };
#pragma unroll
        for (int q = 0; q < S8_CACHE; ++q)
            if (q < nq_max && __ballot(!found) != 0ull) scan(q, wreg[q]);
        for (int q = S8_CACHE; q < nq_max; ++q) {
            if (__ballot(!found) == 0ull) break;
            scan(q, weight(8 * q + gl));
        }
        int parent = -1;
        if (n > 0 && kk < n - 1) parent = i - 1 - kk;
        if (live && gl == 0) {
            const int pnode = parent >= 0 ? a.nodes[parent] : -1;
            if (parents) parents[i] = (int64_t)parent + 1;     // 1-based event index, 0 = baseline
            if (pnodes) pnodes[i] = (int64_t)pnode + 1;
            const int kpos = a.wpos[it.kbeg + kq];              // the child's bucket position
            pn_b[kpos] = pnode;
            dt_b[kpos] = parent >= 0 ? t - a.times[parent] : 0.0;
        }
    }
}

// ---- statistics: workgroup c scans the children of c, 64 at a time (one per lane, time order); wave w owns the
// cells of the parent nodes with (p >> 6) mod 4 == w.  The lanes of a group update their cells in LDS in parallel;
// lanes of one group that hit the SAME cell (about two pairs per group at N = 1024) are serialised lowest lane first
// by a ticket per cell (LDS atomic min), so every (p,c) cell still accumulates in child time order -- the order of the
// reference's serial loops, hence the same bits -- without the per-child serial visit of the first version (82 us).
__global__ __launch_bounds__(NHP_BLOCK) void k_stats(nhp_cont_args a, const int32_t *__restrict__ pn_b,
                                                     const double *__restrict__ dt_b,
                                                     double *__restrict__ cnt0, double *__restrict__ Mn,
                                                     double *__restrict__ Mnm, double *__restrict__ Xnm,
                                                     double *__restrict__ Vnm)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    const int c = a.col_begin + blockIdx.x, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *cnt = reinterpret_cast<double *>(smem);          // [N] events on c attributed to p
    double *sx = cnt + N;                                    // [N] Σ value, then the mean
    double *sv = sx + N;                                     // [N] Σ (value - mean)²
    int *s_pn = reinterpret_cast<int *>(sv + N);             // [NHP_BLOCK] staged parent nodes
    double *s_v = reinterpret_cast<double *>(s_pn + NHP_BLOCK);   // [NHP_BLOCK] staged values
    double *red = s_v + NHP_BLOCK;                           // [NHP_WAVES]
    int *tag = reinterpret_cast<int *>(red + NHP_WAVES);     // [N] lowest pending lane per cell; 64 = free
    const int kb = a.boff[c], ke = a.boff[c + 1];
    const bool lognorm = a.impulse_kind == NHP_IMPULSE_LOGITNORMAL;
    for (int p = tid; p < N; p += NHP_BLOCK) { cnt[p] = 0.0; sx[p] = 0.0; sv[p] = 0.0; tag[p] = 64; }

    double base_cnt = 0.0;
    for (int pass = 0; pass < (lognorm ? 2 : 1); ++pass) {
        for (int k0 = kb; k0 < ke; k0 += NHP_BLOCK) {
            __syncthreads();
            const int k = k0 + tid;
            int pn = -1;
            double v = 0.0;
            if (k < ke) {
                pn = pn_b[k];
                const double d = dt_b[k];
                // log_duration: log((child - parent) / (Δtmax - (child - parent)))  src/impulses.jl:228
                v = (lognorm && pn >= 0) ? nhp_log(d / (a.dt_max - d)) : d;
                if (pass == 0 && pn < 0) base_cnt += 1.0;
            }
            s_pn[tid] = pn;
            s_v[tid] = v;
            __syncthreads();
            const int nb = min(NHP_BLOCK, ke - k0);
            // every wave walks the staged children 64 at a time and takes those whose cell it owns
            for (int ch = 0; ch < nb; ch += 64) {
                const int e = ch + lane;
                const int pe = e < nb ? s_pn[e] : -1;
                const double ve = e < nb ? s_v[e] : 0.0;
                bool pending = pe >= 0 && ((pe >> 6) & (NHP_WAVES - 1)) == wave;
                // LDS operations of one wave execute in program order, and a cell's ticket is only ever touched by
                // its owner wave, so min -> read -> update -> release needs no barrier
                while (__ballot(pending)) {
                    if (pending) (void)__hip_atomic_fetch_min(&tag[pe], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pending && __hip_atomic_load(&tag[pe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == lane) {
                        if (pass == 0) { cnt[pe] += 1.0; sx[pe] = sx[pe] + ve; }
                        else { const double dlt = ve - sx[pe]; sv[pe] = sv[pe] + dlt * dlt; }
                        __hip_atomic_store(&tag[pe], 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        pending = false;
                    }
                }
            }
        }
        if (pass == 0) {
            __syncthreads();
            for (int p = tid; p < N; p += NHP_BLOCK) sx[p] = sx[p] / cnt[p];     // mean; NaN when cnt == 0
        }
    }
    __syncthreads();
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        if (Mnm) Mnm[k] = cnt[p];
        // exponential: fillna!(Xnm ./ Mnm, 0) (src/impulses.jl:95); logit-normal keeps NaN (:222)
        if (Xnm) Xnm[k] = (!lognorm && cnt[p] == 0.0) ? 0.0 : sx[p];
        if (Vnm && lognorm) Vnm[k] = sv[p];
    }
    // baseline-attributed events on c and events on c
    const double b = nhp_block_sum(base_cnt, red);
    if (tid == 0) {
        if (cnt0) cnt0[c] = b;
        if (Mn) Mn[c] = (double)(ke - kb);
    }
}

// Device-resident outputs of one sampler + statistics pass (all inside ctx->d_scratch).
struct samp_out {
    int64_t *parents, *pnodes;
    int32_t *pn_b;
    double *dt_b;
    double *cnt0, *Mn, *Mnm, *X, *V;
};

static nhp_status run_sampler(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const double *u,
                              uint64_t seed, uint64_t step, bool want_parents, bool want_stats, samp_out *o,
                              nhp_status *deferred = nullptr)      // non-null: do not synchronise; *deferred = the PREVIOUS sweep's verdict
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)ds->M, N = (size_t)ds->N, NN = N * N, Mp = M ? M : 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t o_par = carve(8 * Mp), o_pno = carve(8 * Mp), o_dtb = carve(8 * Mp);
    const size_t o_u = carve(u ? 8 * Mp : 8), o_err = carve(8);
    const size_t o_cnt0 = carve(8 * N), o_Mn = carve(8 * N), o_Mnm = carve(8 * NN), o_X = carve(8 * NN), o_V = carve(8 * NN);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, off));
    char *base = (char *)ctx->d_scratch;
    o->parents = (int64_t *)(base + o_par); o->pnodes = (int64_t *)(base + o_pno);
    o->pn_b = ds->d_pn;                      // kept with the dataset: the LGCP baseline update reads it later
    o->dt_b = (double *)(base + o_dtb);
    o->cnt0 = (double *)(base + o_cnt0); o->Mn = (double *)(base + o_Mn); o->Mnm = (double *)(base + o_Mnm);
    o->X = (double *)(base + o_X); o->V = (double *)(base + o_V);
    double *d_u = u ? (double *)(base + o_u) : nullptr;
    (void)o_err;
    int *d_err = ctx->d_err;                 // the context's own word: the deferred check outlives the scratch layout
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), st));
    if (u && M) NHP_HIP(ctx, hipMemcpyAsync(d_u, u, 8 * M, hipMemcpyHostToDevice, st));
    if (nhp_is_column_shard(ds)) {
        // a shard samples the parents of the children on its own nodes; everything it does not own reads as "no parent",
        // and the statistics of the other columns as zero
        if (want_parents) { NHP_HIP(ctx, hipMemsetAsync(o->parents, 0, 8 * Mp, st)); NHP_HIP(ctx, hipMemsetAsync(o->pnodes, 0, 8 * Mp, st)); }
        if (want_stats) NHP_HIP(ctx, hipMemsetAsync(base + o_cnt0, 0, off - o_cnt0, st));
    }

    nhp_cont_args a = nhp_make_args(ds, m);
    const bool expo = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    if (!expo) (void)nhp_ensure_pair_cache(ctx, ds, &a);            // (no cache: the whole pdf per pair, as before)
    // 8 lanes per child (k_sampler8) from a mean window of 24 parents: measured (N=1024, M=1e6, exp | logit-normal)
    // 508 | 846 µs vs 681 | 1228 µs at K̄=64 and 3.3 | 6.3 ms vs 5.0 | 8.9 ms at K̄=512, but 144 | 198 µs vs 115 | 174 µs
    // at K̄=8, where the 8-step chains outweigh the better gathers.  Never when some window reaches Julia's
    // pairwise-sum threshold (> 1024 weights), which only the single-lane kernel implements.
    // NHP_SAMPLER=1 | 8 forces a kernel (8 still respects the threshold).
    const int force = getenv("NHP_SAMPLER") ? atoi(getenv("NHP_SAMPLER")) : 0;
    const double kbar = ds->M > 0 ? (double)ds->pairs / (double)ds->M : 0.0;
    // a sliced dataset (up to 160 pairs per event): one lane per child over the slice planes (cont_slices.hip; same bits) --
    // logit-normal impulses through the planes of logit(x) and 1/(x(1-x)), exponential ones through the plane of exact
    // delays (NHP_SAMPLER_EXPO_SLICES=0: not; measured at K̄ = 8 / 16 / 32 / 64: 54 / 90 / 175 / 346 µs against 80 / 118 / 269 / 488)
    bool sliced = false;
    const int es = getenv("NHP_SAMPLER_EXPO_SLICES") ? atoi(getenv("NHP_SAMPLER_EXPO_SLICES")) : 1;
    if (!force && (!expo || es != 0)) {
        NHP_TRY(nhp_launch_sampler_slices(ctx, ds, m, d_u, seed, step, want_parents ? o->parents : nullptr,
                                          want_parents ? o->pnodes : nullptr, o->pn_b, o->dt_b, d_err, &sliced));
    }
    if (!sliced) {
    const bool coop = force != 1 && ds->max_window + 1 <= 1024 && (force == 8 || kbar >= 24.0);
    const size_t lds = (expo ? 16 : 24) * N + (coop ? 0 : 8 * (size_t)NHP_BLOCK * SAMP_CLD);
    if (lds > 160 * 1024) { nhp_set_error(ctx, "n_nodes = %d exceeds the 160 KiB LDS column budget", ds->N); return NHP_ENOTIMPL; }
    using samp_fn = void (*)(nhp_cont_args, const double *, uint64_t, uint64_t, int64_t *, int64_t *, int32_t *, double *, int *);
    const samp_fn fn = coop ? (expo ? k_sampler8<NHP_IMPULSE_EXPONENTIAL> : k_sampler8<NHP_IMPULSE_LOGITNORMAL>)
                            : (expo ? k_sampler<NHP_IMPULSE_EXPONENTIAL> : k_sampler<NHP_IMPULSE_LOGITNORMAL>);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fn, dim3((unsigned)ds->n_items), dim3(NHP_BLOCK), lds, st, a, d_u, seed, step,
                       want_parents ? o->parents : nullptr, want_parents ? o->pnodes : nullptr, o->pn_b, o->dt_b, d_err);
    NHP_HIP(ctx, hipGetLastError());
    }
    ds->pn_valid = true;
    if (want_stats) {
        const size_t lds_stats = 8 * (3 * N + NHP_BLOCK + NHP_WAVES) + 4 * NHP_BLOCK + 4 * N + 16;
        if (lds_stats > 160 * 1024) { nhp_set_error(ctx, "statistics: n_nodes = %d exceeds the LDS budget", ds->N); return NHP_ENOTIMPL; }
        if (lds_stats > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stats);
        hipLaunchKernelGGL(k_stats, dim3((unsigned)(ds->col_end - ds->col_begin)), dim3(NHP_BLOCK), lds_stats, st, a, o->pn_b, o->dt_b, o->cnt0, o->Mn, o->Mnm, o->X, o->V);
        NHP_HIP(ctx, hipGetLastError());
    }
    // the flag of the sweep before this one (if it was deferred) is on the host by now: this sweep is enqueued behind it
    const nhp_status earlier = nhp_check_deferred(ctx);
    NHP_HIP(ctx, hipMemcpyAsync(ctx->h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    if (deferred) {
        NHP_HIP(ctx, hipEventRecord(ctx->ev_err, st));
        ctx->err_pending = true;
        *deferred = earlier;
        return NHP_OK;
    }
    NHP_HIP(ctx, hipStreamSynchronize(st));
    if (earlier != NHP_OK) return earlier;
    if (*ctx->h_err) {
        *ctx->h_err = 0;
        nhp_set_error(ctx, "resample_parents: weights of some event do not sum to a positive finite value");
        return NHP_EDOMAIN;
    }
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_resample_parents(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                                const nhp_cont_model *m, const double *u,
                                                uint64_t seed, uint64_t step,
                                                int64_t *parents, int64_t *parentnodes, nhp_cont_stats *stats)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    samp_out o;
    NHP_TRY(run_sampler(ctx, ds, m, u, seed, step, parents || parentnodes, stats != nullptr, &o));
    const size_t M = (size_t)ds->M, N = (size_t)ds->N, NN = N * N;
    hipStream_t st = ctx->stream;
    (void)st;
    if (stats) {                                   // through the pinned staging buffer (nhp_download synchronises)
        if (stats->cnt0) NHP_TRY(nhp_download(ctx, stats->cnt0, o.cnt0, 8 * N));
        if (stats->Mn) NHP_TRY(nhp_download(ctx, stats->Mn, o.Mn, 8 * N));
        if (stats->Mnm) NHP_TRY(nhp_download(ctx, stats->Mnm, o.Mnm, 8 * NN));
        if (stats->Xnm) NHP_TRY(nhp_download(ctx, stats->Xnm, o.X, 8 * NN));
        if (stats->Vnm && m->impulse_kind == NHP_IMPULSE_LOGITNORMAL) NHP_TRY(nhp_download(ctx, stats->Vnm, o.V, 8 * NN));
    }
    if (parents && M) NHP_TRY(nhp_download(ctx, parents, o.parents, 8 * M));
    if (parentnodes && M) NHP_TRY(nhp_download(ctx, parentnodes, o.pnodes, 8 * M));
    return NHP_OK;
}

// ---- device-side conjugate draws: the counter-based Gamma / Normal generators live in nhp_rng.h
struct gibbs_priors { double alpha0, beta0, kappa, nu, a, b, mu_mu, kappa_mu; };

// A column shard draws the parameters of its own columns (λ0_c and column c of W, θ | μ, τ); Mn[p], the events on the
// PARENT node, comes from the dataset's counts, which a shard has for every node.
__global__ __launch_bounds__(256) void k_gibbs_draw(int N, int col_begin, int col_end, int impulse_kind, double duration, gibbs_priors pr,
                                                    uint64_t seed, uint64_t step,
                                                    const double *__restrict__ cnt0, const double *__restrict__ Mn,
                                                    const double *__restrict__ Mnm, const double *__restrict__ X,
                                                    const double *__restrict__ V, double *__restrict__ lambda0,
                                                    double *__restrict__ p1, double *__restrict__ p2, double *__restrict__ W)
{
    const size_t NN = (size_t)N * N;
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < (size_t)N && (int)k >= col_begin && (int)k < col_end)      // λ0_c ~ Gamma(α0 + cnt0_c, 1/(β0 + T))
        lambda0[k] = dev_gamma(pr.alpha0 + cnt0[k], rng_rcp(pr.beta0 + duration), seed ^ 0x243F6A8885A308D3ull, step, k);
    if (k >= NN) return;
    const int col = (int)((uint32_t)k / (uint32_t)N);
    if (col < col_begin || col >= col_end) return;
    const double m = Mnm[k];
    // W[p,c] ~ Gamma(κ + Mnm, 1/(ν + Mn[p]))
    W[k] = dev_gamma(pr.kappa + m, rng_rcp(pr.nu + Mn[(uint32_t)k % (uint32_t)N]), seed ^ 0x13198A2E03707344ull, step, k);
    if (impulse_kind == NHP_IMPULSE_EXPONENTIAL) {
        // θ ~ Gamma(α + Mnm, 1/(β + Mnm·Xnm))      (Xnm = 0 where Mnm = 0)
        p1[k] = dev_gamma(pr.a + m, rng_rcp(pr.b + m * X[k]), seed ^ 0xA4093822299F31D0ull, step, k);
    } else {
        // τ ~ Gamma(α0 + Mnm/2, 1/βnm),  μ ~ Normal(μnm, ((κμ + Mnm)τ)^-½); NaN (no observations) -> prior
        const double x = X[k];
        const double rkm = rng_rcp(m + pr.kappa_mu);
        double bnm = 0.5 * V[k] + m * pr.kappa_mu * rkm * (x - pr.mu_mu) * (x - pr.mu_mu) * 0.5;
        if (bnm != bnm) bnm = pr.b;
        double mnm = (pr.kappa_mu * pr.mu_mu + m * x) * rkm;
        if (mnm != mnm) mnm = pr.mu_mu;
        const double tau = dev_gamma(pr.a + 0.5 * m, rng_rcp(bnm), seed ^ 0x082EFA98EC4E6C89ull, step, k);
        p2[k] = tau;
        p1[k] = mnm + dev_normal(seed ^ 0x452821E638D01377ull, step, k, 0) * rng_rsqrt((pr.kappa_mu + m) * tau);
    }
}

extern "C" nhp_status nhp_cont_gibbs_step(nhp_ctx *ctx, const nhp_cont_dataset *ds, nhp_cont_model *m,
                                          const nhp_gibbs_priors *pr, uint64_t seed, uint64_t step)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (!pr) return NHP_EINVAL;
    if (m->baseline_kind != NHP_BASELINE_HOMOGENEOUS) { nhp_set_error(ctx, "gibbs_step: homogeneous baseline only"); return NHP_ENOTIMPL; }
    samp_out o;
    // no synchronisation inside a sweep: its error flag is looked at one sweep later (nhp_check_deferred)
    nhp_status earlier = NHP_OK;
    NHP_TRY(run_sampler(ctx, ds, m, nullptr, seed, step, false, true, &o, &earlier));
    gibbs_priors g{pr->alpha0, pr->beta0, pr->kappa, pr->nu, pr->a, pr->b, pr->mu_mu, pr->kappa_mu};
    const size_t NN = (size_t)ds->N * ds->N;
    ++m->version;
    hipLaunchKernelGGL(k_gibbs_draw, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, ctx->stream, ds->N, ds->col_begin, ds->col_end,
                       m->impulse_kind, ds->duration, g, seed, step, o.cnt0, ds->d_cnt, o.Mnm, o.X, o.V, m->d_lambda0, m->d_p1, m->d_p2, m->d_W);
    NHP_HIP(ctx, hipGetLastError());
    return earlier;
}

// ---- sample store: running first and second moments of the chain on the device.  mcmc! keeps
// params(process) of every step (src/inference.jl:61) -- 4N²+N doubles, 33.5 MB per step at N = 1024, i.e. more PCIe
// time than the whole sweep takes -- while what a chain is read for are posterior means and variances (chains.py
// gathers exactly these).  Order: params(process) of the standard process, then vec(A) when the model has one.
__global__ __launch_bounds__(256) void k_moments(int64_t N, int64_t nimp, const double *__restrict__ lambda0,
                                                 const double *__restrict__ p1, const double *__restrict__ p2,
                                                 const double *__restrict__ W, const double *__restrict__ A,
                                                 int64_t len, double *__restrict__ mom, double *__restrict__ rho)
{
    const int64_t NN = N * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
        double x;
        if (i < N) x = lambda0[i];
        else if (i < N + NN) x = p1[i - N];
        else if (i < N + nimp) x = p2[i - N - NN];
        else if (i < N + nimp + NN) x = W[i - N - nimp];
        else x = A[i - N - nimp - NN];
        mom[i] += x;
        mom[len + i] += x * x;
    }
    if (rho && blockIdx.x == 0 && threadIdx.x == 0) { rho[1] += rho[0]; rho[2] += rho[0] * rho[0]; }
}

extern "C" nhp_status nhp_cont_model_moments_reset(nhp_ctx *ctx, nhp_cont_model *m)
{
    if (!ctx || !m) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (m->baseline_kind != NHP_BASELINE_HOMOGENEOUS) { nhp_set_error(ctx, "moments: homogeneous baseline only"); return NHP_ENOTIMPL; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t N = m->N, NN = N * N;
    const int64_t len = N + (m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN) + NN + (m->has_A ? NN : 0);
    if (!m->d_mom || m->mom_len != len) {
        if (m->d_mom) { NHP_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(m->d_mom); m->d_mom = nullptr; }
        if (hipMalloc((void **)&m->d_mom, sizeof(double) * 2 * (size_t)len) != hipSuccess) {
            nhp_set_error(ctx, "out of device memory (sample moments)");
            return NHP_ENOMEM;
        }
        m->mom_len = len;
    }
    NHP_HIP(ctx, hipMemsetAsync(m->d_mom, 0, sizeof(double) * 2 * (size_t)len, ctx->stream));
    if (m->d_rho) NHP_HIP(ctx, hipMemsetAsync(m->d_rho + 1, 0, 2 * sizeof(double), ctx->stream));
    m->mom_count = 0;
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_moments_accumulate(nhp_ctx *ctx, nhp_cont_model *m)
{
    if (!ctx || !m) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!m->d_mom) NHP_TRY(nhp_cont_model_moments_reset(ctx, m));
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t N = m->N, NN = N * N, nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    const unsigned blocks = (unsigned)std::min<int64_t>((m->mom_len + 255) / 256, 4096);
    hipLaunchKernelGGL(k_moments, dim3(blocks), dim3(256), 0, ctx->stream, N, nimp, m->d_lambda0, m->d_p1, m->d_p2, m->d_W,
                       m->has_A ? m->d_A : nullptr, m->mom_len, m->d_mom, m->d_rho);
    NHP_HIP(ctx, hipGetLastError());
    ++m->mom_count;
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_moments_fetch(nhp_ctx *ctx, const nhp_cont_model *m, double *sum, double *sumsq,
                                                   int64_t len, int64_t *count)
{
    if (!ctx || !m || !sum || !sumsq || !count) return NHP_EINVAL;
    if (!m->d_mom) { nhp_set_error(ctx, "moments: nothing accumulated"); return NHP_EINVAL; }
    if (len != m->mom_len) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_download(ctx, sum, m->d_mom, sizeof(double) * (size_t)len));
    NHP_TRY(nhp_download(ctx, sumsq, m->d_mom + len, sizeof(double) * (size_t)len));
    *count = m->mom_count;
    return NHP_OK;
}

// params(process) of the device-resident model, standard order [λ0; θ | μ; τ; W]
extern "C" nhp_status nhp_cont_model_get_params(nhp_ctx *ctx, const nhp_cont_model *m, double *x, int64_t len)
{
    if (!ctx || !m || !x) return NHP_EINVAL;
    if (m->baseline_kind != NHP_BASELINE_HOMOGENEOUS) return NHP_ENOTIMPL;
    const size_t N = (size_t)m->N, NN = N * N;
    const size_t nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    if ((size_t)len != N + nimp + NN) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    (void)st;
    NHP_TRY(nhp_download(ctx, x, m->d_lambda0, 8 * N));
    NHP_TRY(nhp_download(ctx, x + N, m->d_p1, 8 * NN));
    if (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL) NHP_TRY(nhp_download(ctx, x + N + NN, m->d_p2, 8 * NN));
    NHP_TRY(nhp_download(ctx, x + N + nimp, m->d_W, 8 * NN));
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_model_get_adjacency(nhp_ctx *ctx, const nhp_cont_model *m, double *A, int64_t len)
{
    if (!ctx || !m || !A) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!m->has_A) { nhp_set_error(ctx, "get_adjacency: the model has no adjacency matrix"); return NHP_EINVAL; }
    const size_t NN = (size_t)m->N * (size_t)m->N;
    if ((size_t)len != NN) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    return nhp_download(ctx, A, m->d_A, 8 * NN);
}


// ---- diagnostics: the device-side random variates themselves (tests hold them to their distributions and to known answers)
// kind 0: Gamma(shape a[i], scale b[i]); kind 1: standard normal; kind 2: Beta(a[i], b[i]) as X/(X+Y) -- each with the
// generators and the Philox keying (seed, step, element i | 2i, 2i+1) the Gibbs kernels use.
__global__ __launch_bounds__(256) void k_probe_draws(int kind, uint64_t seed, uint64_t step, int64_t n, const double *__restrict__ pa,
                                                     const double *__restrict__ pb, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (kind == 0) out[i] = dev_gamma(pa[i], pb[i], seed, step, (uint64_t)i);
    else if (kind == 1) out[i] = dev_normal(seed, step, (uint64_t)i, 0);
    else {
        const double x = dev_gamma(pa[i], 1.0, seed, step, (uint64_t)(2 * i)), y = dev_gamma(pb[i], 1.0, seed, step, (uint64_t)(2 * i + 1));
        out[i] = x / (x + y);
    }
}

extern "C" nhp_status nhp_probe_draws(nhp_ctx *ctx, int32_t kind, uint64_t seed, uint64_t step, int64_t n, const double *a,
                                      const double *b, double *out)
{
    if (!ctx || !out || n < 0 || kind < 0 || kind > 2 || (kind != 1 && (!a || !b))) return NHP_EINVAL;
    if (n == 0) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 3 * sizeof(double) * (size_t)n));
    double *da = (double *)ctx->d_scratch, *db = da + n, *dout = db + n;
    if (kind != 1) {
        NHP_HIP(ctx, hipMemcpyAsync(da, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        NHP_HIP(ctx, hipMemcpyAsync(db, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(k_probe_draws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, kind, seed, step, n, da, db, dout);
    NHP_HIP(ctx, hipGetLastError());
    return nhp_download(ctx, out, dout, sizeof(double) * (size_t)n);
}
