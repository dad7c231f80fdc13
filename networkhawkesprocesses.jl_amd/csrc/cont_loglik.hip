// Windowed continuous log-likelihood: Σ_i log λ_{c_i}(t_i) with
//   λ_c(t_i) = λ0_c(t_i) + Σ_{j<i, t_j > t_i-Δtmax} A[n_j,c] W[n_j,c] ħ(t_i - t_j; θ[n_j,c])
// (reference: loglikelihood src/continuous.jl:210-239,360-389; total_intensity :286-300,
//  :391-405; impulse_response :302-305,:521-525).
//
// Mapping to CDNA4
//  * Child events are bucketed by node (done once per dataset).  A workgroup owns a run of
//    children of ONE node c, so column c of the parameter tables -- contiguous in the
//    reference's column-major layout -- is staged once in LDS (16-24 B per parent node) and
//    every W[n_j,c] / θ[n_j,c] gather becomes an LDS read instead of a random HBM/L2 access
//    into an N x N table.
//  * G lanes (a power of two chosen from the mean window length) cooperate on one child:
//    lane g reads parents i-1-g, i-1-g-G, ... so a group's loads of (times, nodes) are
//    contiguous, and the group sum is a log2(G)-step cross-lane reduction.
//  * The per-event row-sum loop of the reference (Σ_i Σ_c W[n_i,c], :219-221) is regrouped
//    as Σ_c Σ_p cnt[p] W[p,c] and folded into the column staging of the node's first item,
//    so every parameter is read from HBM exactly once per evaluation.
//  * Sums are reduced lane -> wave -> block -> one partial per workgroup; a one-block second
//    kernel adds the partials in a fixed order (deterministic, no atomics).
#include <utility>
#include <stdio.h>

#include "nhp_internal.h"
#include "nhp_math.h"

// ---- cross-lane sum over groups of G lanes.  Offsets < 16 stay in the VALU through DPP
// (quad_perm / row_half_mirror / row_mirror); 16 and 32 go through ds_bpermute.
template <int G>
__device__ __forceinline__ double group_sum(double v)
{
    if (G >= 2) v = nhp_dpp_add(v, 0);
    if (G >= 4) v = nhp_dpp_add(v, 1);
    if (G >= 8) v = nhp_dpp_add(v, 2);
    if (G >= 16) v = nhp_dpp_add(v, 3);
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Baseline intensity λ0_c(t): HomogeneousProcess (src/baselines.jl:115-118) or the
// LogGaussianCoxProcess evaluator = LinearInterpolator on the grid
// (src/baselines.jl:332-334, src/utils/interpolation.jl:27-36; bin i iff x[i] <= t < x[i+1],
// y[end] at the right edge).  Binary search finds the same bin as the reference's scan.
__device__ __forceinline__ double baseline_at(const nhp_cont_args &a, int c, double t)
{
#pragma clang fp contract(off)
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return a.lambda0[c];
    const double *x = a.grid;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    int lo = 0, hi = a.grid_n - 1;          // invariant: x[lo] <= t, and t < x[hi] or hi is the end
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

// -Σ_c ∫λ0_c: λ .* duration (src/baselines.jl:98-102) or the trapezoid rule over the grid, which
// ignores `duration` (src/baselines.jl:336, src/utils/interpolation.jl:40-48).  Per-thread part.
__device__ __forceinline__ double baseline_integral_part(const nhp_cont_args &a)
{
    double sb = 0.0;
    for (int c = a.col_begin + threadIdx.x; c < a.col_end; c += blockDim.x) {
        if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) {
            sb += a.lambda0[c] * a.duration;
        } else {
            const double *y = a.lambda0 + (size_t)c * a.grid_n;
            double I = 0.0;
            for (int i = 0; i + 1 < a.grid_n; ++i) I += 0.5 * (y[i] + y[i + 1]) * (a.grid[i + 1] - a.grid[i]);
            sb += I;
        }
    }
    return sb;
}

// One column's share of the same, spread over the threads of the workgroup (their block sum adds it up).
__device__ __forceinline__ double baseline_integral_col(const nhp_cont_args &a, int c)
{
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return threadIdx.x == 0 ? a.lambda0[c] * a.duration : 0.0;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    double I = 0.0;
    for (int i = threadIdx.x; i + 1 < a.grid_n; i += blockDim.x) I += 0.5 * (y[i] + y[i + 1]) * (a.grid[i + 1] - a.grid[i]);
    return I;
}

#ifdef NHP_STAMP      // diagnostic build only (tools/stamps.py): s_memtime at the phase boundaries of wave 0 of every workgroup
__device__ unsigned long long g_stamps[8 * 4096];
#define NHP_STAMP_AT(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int nhp_debug_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)n);
}
#else
#define NHP_STAMP_AT(i) do { } while (0)
#endif

// U children per group are in flight at once: their child records, then their parents'
// packed (t, node) records, are fetched by independent loads before any is consumed, so the
// dependent global-load chains of different children overlap.  Inactive slots are predicated
// (clamped address, masked accumulate) rather than branched, which is what lets the loads batch.
#define NHP_SHARDS 64

// PACK: the parents come as 8-byte records (node << 48 | fixed-point time, nhp_cont_dataset::d_ev8) instead of 16-byte
// {t, node} -- half the bytes of the scattered window fetches that bound the short-window kernel.  The time field is decoded
// with the 2^52 trick (the 48 bits OR-ed under the exponent of 2^52 ARE the double 2^52 + q; one exact subtraction), Δt =
// ((t_child - t0)·2^s - q)·2^-s: the only error is the parent's rounding to 2^-s, below 2^-49 of the data's span (log-
// likelihood of the metric data: 4·10^-14 relative; the window bounds come from the exact times either way).  Exponential
// impulses without the λ output only: a logit-normal Δt/Δtmax must stay strictly inside (0, 1).
template <int IMP, int G, int U, bool PACK = false>
__global__ __launch_bounds__(NHP_WBLOCK) void k_windowed(nhp_cont_args a, int mask_integral,
                                                        double *__restrict__ partials,
                                                        double *__restrict__ lambda_out,
                                                        unsigned int *__restrict__ counter,
                                                        double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [waves <= 16] + flag at [16]
    double2 *col = reinterpret_cast<double2 *>(smem + 192);         // [N]
    double *colw = reinterpret_cast<double *>(col + a.N);           // [N], logit-normal only
    double *etab = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : a.N);   // [64] 2^(j/64) for nhp_exp_neg_tab
    nhp_exp_tab_init(etab);

    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x;
    NHP_STAMP_AT(0);

    // ---- stage column c; the node's first item also owns the column's integral term
    double integ = 0.0;
    for (int p = tid; p < (NHP_SKIP(a, 2) ? 0 : N); p += NHP_WBLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        double w = a.W[k], wint = w;
        if (a.A) {
            w = a.A[k] * w;
            if (mask_integral) wint = w;
        }
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            const double th = a.p1[k];
            // term = (a·w·θ)·exp(-θ·Δt): the rate times 64/ln 2 (nhp_exp_neg_tab_scaled) and, for packed records, times the
            // time unit of their delays
            col[p] = make_double2(-(th * 92.33248261689366) * (PACK ? a.ev8_inv : 1.0), w * th);
        } else {
            col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
            colw[p] = w;
        }
        if (it.first) integ += a.cnt[p] * wint;
    }
    // the node's first item also carries the column's baseline integral (fused second stage only: the two-kernel path
    // leaves it to k_finalize), so the last workgroup's tail has nothing to load but the partial sums
    if (out && it.first && !NHP_SKIP(a, 2)) integ += baseline_integral_col(a, c);
    __syncthreads();
    NHP_STAMP_AT(1);

    // ---- children: G lanes per child, U children per group in flight
    constexpr int GROUPS = NHP_WBLOCK / G;
    const int gid = tid / G, gl = tid % G;
    const int nchild = it.kend - it.kbeg;
    static_assert(!PACK || IMP == NHP_IMPULSE_EXPONENTIAL, "packed records: exponential impulses only");
    struct rec { nhp_event e; unsigned long long w; };             // (one of the two is used; the other is never loaded)
    auto fetch = [&](const int jj) {
        rec r;
        if (PACK) r.w = a.ev8[jj > 0 ? jj : 0];
        else r.e = a.ev[jj > 0 ? jj : 0];
        return r;
    };
    auto node_of = [&](const rec &r) { return PACK ? (int)(r.w >> 48) : r.e.node; };
    // Δt of a parent: tc = the child's time, exact, or (t_child - t0)·2^s for packed records
    auto delay = [&](const double tc, const rec &r) {
        if (!PACK) return tc - r.e.t;
        const double d = __hiloint2double((int)(((unsigned)(r.w >> 32) & 0xFFFFu) | 0x43300000u), (int)(unsigned)r.w);
        return tc - (d - 4503599627370496.0);             // in units of 2^-s: the column's rate carries the factor
    };
    // slot of (wave, u, group-in-wave) inside a round: a wave's U*GW children are contiguous in the
    // round's window-sorted order
    constexpr int GW = 64 / G;
    const int slot0 = (gid / GW) * (GW * U) + (gid % GW);
    // Σ log λ as one logarithm per lane: log Π λ_k = log(Π mant_k) + ln2·Σ exp_k (see k_windowed_batch); only a group's
    // first lane holds a λ, so the old scheme parked the λ_k in LDS and took the ~45-instruction logs in a second pass
    double prod = 1.0;
    int pexp = 0;
    for (int r0 = 0; r0 < (NHP_SKIP(a, 1) ? 0 : nchild); r0 += GROUPS * U) {
        double t[U], tc[U], s[U];
        int j[U], f[U], idx[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = r0 + slot0 + u * GW;
            valid[u] = kk < nchild;
            const nhp_child ch = a.child_w[it.kbeg + (valid[u] ? kk : r0)];
            t[u] = ch.t; idx[u] = ch.idx;
            tc[u] = PACK ? (ch.t - a.ev8_t0) * a.ev8_scale : ch.t;
            j[u] = ch.idx - 1 - gl;
            f[u] = valid[u] ? ch.first : 0x7fffffff;
            s[u] = 0.0;
        }
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) more |= j[u] >= f[u];
        if (NHP_SKIP(a, 8)) more = false;
        rec e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = fetch(j[u]);
        // steady state (wide groups = long windows only; measured slower for G <= 16): iterations in
        // which every one of this lane's U slots still has a parent, so nothing is predicated ...
        int nfull = G >= 32 ? 0x7fffffff : 0;
        if (G >= 32) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int left = j[u] >= f[u] ? (j[u] - f[u]) / G + 1 : 0;
                nfull = left < nfull ? left : nfull;
            }
        }
        if (NHP_SKIP(a, 8)) nfull = 0;
        for (int itn = 0; itn < nfull; ++itn) {
            rec en[U];
#pragma unroll
            for (int u = 0; u < U; ++u) en[u] = fetch(j[u] - G);
            asm volatile("" ::: "memory");          // keeps the prefetch a prefetch (see k_windowed_batch)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double dt = delay(tc[u], e[u]);
                const double2 q = col[node_of(e[u])];
                if (IMP == NHP_IMPULSE_EXPONENTIAL) s[u] += q.y * nhp_exp_neg_tab_scaled(q.x * dt, etab);
                else s[u] += colw[node_of(e[u])] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
                j[u] -= G;
                e[u] = en[u];
            }
        }
        // ... then the ragged tail, predicated per slot
        more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) more |= j[u] >= f[u];
        if (NHP_SKIP(a, 8)) more = false;
        while (more) {
            rec en[U];                             // next iteration's parents, in flight under the math
#pragma unroll
            for (int u = 0; u < U; ++u) en[u] = fetch(j[u] - G);
            // Wide groups (long windows): keep the prefetch a prefetch (see k_windowed_batch).  Narrow groups run one or two
            // trips per child: there the compiler's folding of the prefetch into the consuming trip SAVES the load a last
            // trip would waste (K = 8: 41.5 us folded, 52.7 us with the barrier).
            if (G >= 16) asm volatile("" ::: "memory");
            more = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double dt = delay(tc[u], e[u]);
                const double2 q = col[node_of(e[u])];
                double term;
                if (IMP == NHP_IMPULSE_EXPONENTIAL) term = q.y * nhp_exp_neg_tab_scaled(q.x * dt, etab);
                else term = colw[node_of(e[u])] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
                s[u] += (j[u] >= f[u]) ? term : 0.0;
                j[u] -= G;
                more |= j[u] >= f[u];
                e[u] = en[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = group_sum<G>(s[u]);
            if (gl == 0 && valid[u]) {
                const double lam = baseline_at(a, c, t[u]) + s[u];
                if (lambda_out) lambda_out[idx[u]] = lam;
                prod *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
                pexp += __builtin_amdgcn_frexp_exp(lam);
            }
        }
        pexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
    }
    NHP_STAMP_AT(2);
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;
    if (prod == 0.0) acc = -__builtin_inf();
    static_assert(2 * (NHP_WBLOCK / 64) <= 16, "red[] holds 16 doubles ahead of the flag");
    double blk = acc, blk_int = integ;
    nhp_block_sum2_n<NHP_WBLOCK / 64>(blk, blk_int, red);
    if (!out) {
        if (tid == 0) {
            partials[2 * (size_t)blockIdx.x] = blk;
            partials[2 * (size_t)blockIdx.x + 1] = blk_int;
        }
        return;
    }
    // ---- fused second stage: the workgroup that draws the last ticket adds all partials in a
    // fixed order (deterministic).  Hand-off per the sc1 recipe (cdna_hip_programming.md G16):
    // write-through stores of the partials, drain, one relaxed agent-scope ticket; the reader
    // uses sc1 loads only.  The last workgroup leaves the ticket counter at 0 for the next launch.
    int *flag = reinterpret_cast<int *>(red + 16);
    if (tid == 0) {
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // One ticket word saturates near 88 atomics/us (MI355X_MICROARCH "dequeue"), i.e. >20 us for
        // 2048 workgroups, so tickets are sharded over NHP_SHARDS words (one 128-B line each); the
        // last arriver of a shard draws a ticket on the top word.
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SHARDS - 1) / NHP_SHARDS;          // workgroups in this shard
        const unsigned int used = nb < NHP_SHARDS ? nb : NHP_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    NHP_STAMP_AT(3);
    NHP_STAMP_AT(4);
    if (!*flag) return;
    double sl = 0.0, si = 0.0;
    for (unsigned int i = tid; i < gridDim.x; i += NHP_WBLOCK) {
        sl += __hip_atomic_load(&partials[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si += __hip_atomic_load(&partials[2 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nhp_block_sum2_n<NHP_WBLOCK / 64>(sl, si, red);
    if (tid == 0) {
        *out = (0.0 - si) + sl;                  // si: baseline integrals + Σ cnt·w, column by column
    }
    if (tid <= NHP_SHARDS) __hip_atomic_store(&counter[32 * tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- short windows through the cached pair list --------------------------------------------------------------
//   At K̄ ≈ 8 a launch of k_windowed spends its time waiting: per round of children three DEPENDENT global
// round trips (child records -> first parents -> second parents) to ~10⁶ scattered 64-byte windows.  Which pairs exist
// and their delays are data, not parameters: the dataset keeps them as a list, child by child in the order the items
// visit them, 8 bytes a pair -- parent node << 48 | Δt as a 48-bit fraction of Δtmax (resolution 2⁻⁴⁹·Δtmax: finer than
// the fp64 subtraction t_child - t_parent itself resolves at t ~ 10⁵) -- built once on the device (k_pairs_build) at the
// first evaluation.  A workgroup then stages its item's pair OFFSETS with the column (4 bytes a child) and every round is
// ONE round trip to a contiguous run of the list: 8·P + 4·M bytes per evaluation, streamed, against 133 MB of scattered
// sectors.  Same sum in the same order as k_windowed (lane g of a child's group takes its pairs g, g+G, ..., most recent
// parent first; the same group reduction), exponential impulses, no λ output.
__global__ __launch_bounds__(256) void k_pairs_build(nhp_cont_args a, uint64_t *__restrict__ plist)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.M) return;
    const nhp_child ch = a.child_w[k];
    const uint32_t o = a.poff[k];
    const double scale = a.inv_dtmax * 281474976710656.0;          // 2^48 / Δtmax
    for (int r = 0; r < ch.idx - ch.first; ++r) {
        const nhp_event e = a.ev[ch.idx - 1 - r];
        double q = __builtin_rint((ch.t - e.t) * scale);
        q = q < 1.0 ? 1.0 : (q > 281474976710655.0 ? 281474976710655.0 : q);
        plist[(size_t)o + r] = ((uint64_t)(uint32_t)e.node << 48) | (uint64_t)q;
    }
}

// IMP = logit-normal reads the sampler's pair cache instead ({logit(x), 1/(x(1-x))} + node, 18 bytes a pair): the pdf's
// logarithm and division are data (nhp_pdf_logitnormal_cached), what is left per pair is one exponential.
template <int IMP, int G, int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_windowed_pairs(nhp_cont_args a, int mask_integral, int max_item,
                                                              double *__restrict__ partials, double *__restrict__ lambda_out,
                                                              unsigned int *__restrict__ counter,
                                                              double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2 * waves <= 32] + flag at [32]
    double2 *col = reinterpret_cast<double2 *>(smem + 320);         // [N] exp: {θ·Δtmax·2⁻⁴⁸, a·w·θ} (the delay arrives as an integer); logit: {μ, √τ}
    double *colw = reinterpret_cast<double *>(col + a.N);           // [N] logit-normal: a·w
    double *etab = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : a.N);   // [64] 2^(j/64) for nhp_exp_neg_tab
    uint32_t *off = reinterpret_cast<uint32_t *>(etab + 64);        // [max_item + 1] pair offsets of the item's children
    nhp_exp_tab_init(etab);

    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x;
    NHP_STAMP_AT(0);
    const int nchild = it.kend - it.kbeg;
    const uint32_t base = a.poff[it.kbeg];
    const double unit = a.dt_max * 3.5527136788005009e-15;         // Δtmax · 2⁻⁴⁸
    double integ = 0.0;
    for (int p = tid; p < N; p += BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        double w = a.W[k], wint = w;
        if (a.A) {
            w = a.A[k] * w;
            if (mask_integral) wint = w;
        }
        const double th = a.p1[k];
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            col[p] = make_double2(-((th * unit) * 92.33248261689366), w * th);   // term = (a·w·θ)·exp(-(θ·unit)·q), the rate times 64/ln 2
        } else {
            col[p] = make_double2(th, __builtin_sqrt(a.p2[k]));
            colw[p] = w;
        }
        if (it.first) integ += a.cnt[p] * wint;
    }
    for (int i = tid; i <= nchild; i += BLOCK) off[i] = a.poff[it.kbeg + i] - base;
    if (out && it.first) integ += baseline_integral_col(a, c);
    __syncthreads();
    NHP_STAMP_AT(1);

    constexpr int GROUPS = BLOCK / G, GW = 64 / G;
    const int gid = tid / G, gl = tid % G;
    const int slot0 = (gid / GW) * (GW * U) + (gid % GW);
    const uint64_t *pl = a.plist + (IMP == NHP_IMPULSE_EXPONENTIAL ? base : 0);
    const double2 *plq = a.plq + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : base);
    const uint16_t *pnd = a.pnode + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : base);
    const bool flat = a.baseline_kind == NHP_BASELINE_HOMOGENEOUS;
    const double lam0 = flat ? a.lambda0[c] : 0.0;
    double prod = 1.0;
    int pexp = 0;
    struct rec { uint64_t w; double2 d; int p; };                  // (exp: w; logit-normal: d, p)
    auto fetch = [&](const int jj) {
        rec r;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) r.w = pl[jj];
        else { r.d = plq[jj]; r.p = pnd[jj]; }
        return r;
    };
    auto term = [&](const rec &r) {
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            const double q = __hiloint2double((int)(((unsigned)(r.w >> 32) & 0xFFFFu) | 0x43300000u), (int)(unsigned)r.w) - 4503599627370496.0;
            const double2 cw = col[(int)(r.w >> 48)];
            return cw.y * nhp_exp_neg_tab_scaled(cw.x * q, etab);
        } else {
            const double2 cw = col[r.p];
            return colw[r.p] * nhp_pdf_logitnormal_cached(cw.x, cw.y, r.d);
        }
    };
    // Every address of every round is known once the offsets are staged, so a round's two trips are requested while the
    // round before it is being summed (two register sets, the loop unrolled by two: no set is ever copied).  Both trips are
    // always requested -- a lane without a second pair re-reads its first, the same sector -- so that the requests in
    // flight are the same on every path.
    struct slot_set { int j[U], e[U]; rec w[U], w2[U]; };
    auto issue = [&](const int r0, slot_set &q) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = r0 + slot0 + u * GW;
            const bool v = kk < nchild;
            q.j[u] = v ? (int)off[kk] + gl : 0;
            q.e[u] = v ? (int)off[kk + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j1 = q.j[u] < q.e[u] ? q.j[u] : 0;
            q.w[u] = fetch(j1);
            q.w2[u] = fetch(q.j[u] + G < q.e[u] ? q.j[u] + G : j1);
        }
    };
    auto consume = [&](const int r0, slot_set &q) {
        double s[U];
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = q.j[u] < q.e[u] ? term(q.w[u]) : 0.0;
            s[u] += q.j[u] + G < q.e[u] ? term(q.w2[u]) : 0.0;
            q.j[u] += 2 * G;
            more |= q.j[u] < q.e[u];
        }
        while (more) {                                              // third and later trips: long windows, rare
            more = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (q.j[u] < q.e[u]) s[u] += term(fetch(q.j[u]));
                q.j[u] += G;
                more |= q.j[u] < q.e[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = group_sum<G>(s[u]);
            const int kk = r0 + slot0 + u * GW;
            if (gl == 0 && kk < nchild) {
                const double lam = (flat ? lam0 : baseline_at(a, c, a.child_w[it.kbeg + kk].t)) + s[u];
                if (lambda_out) lambda_out[a.child_w[it.kbeg + kk].idx] = lam;          // (total_intensity, the gradient's 1/λ)
                prod *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
                pexp += __builtin_amdgcn_frexp_exp(lam);
            }
        }
        pexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
    };
    constexpr int STEP = GROUPS * U;
    slot_set qa, qb;
    issue(0, qa);
    for (int r0 = 0; r0 < nchild; r0 += 2 * STEP) {
        issue(r0 + STEP, qb);
        asm volatile("" ::: "memory");
        consume(r0, qa);
        issue(r0 + 2 * STEP, qa);
        asm volatile("" ::: "memory");
        consume(r0 + STEP, qb);
    }
    NHP_STAMP_AT(2);
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;
    if (prod == 0.0) acc = -__builtin_inf();
    double blk = acc, blk_int = integ;
    nhp_block_sum2_n<BLOCK / 64>(blk, blk_int, red);
    if (!out) {
        if (tid == 0) {
            partials[2 * (size_t)blockIdx.x] = blk;
            partials[2 * (size_t)blockIdx.x + 1] = blk_int;
        }
        return;
    }
    // fused second stage: as k_windowed
    int *flag = reinterpret_cast<int *>(red + 32);
    if (tid == 0) {
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SHARDS - 1) / NHP_SHARDS;
        const unsigned int used = nb < NHP_SHARDS ? nb : NHP_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    NHP_STAMP_AT(3);
    NHP_STAMP_AT(4);
    if (!*flag) return;
    double sl = 0.0, si = 0.0;
    for (unsigned int i = tid; i < gridDim.x; i += BLOCK) {
        sl += __hip_atomic_load(&partials[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si += __hip_atomic_load(&partials[2 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nhp_block_sum2_n<BLOCK / 64>(sl, si, red);
    if (tid == 0) *out = (0.0 - si) + sl;
    if (tid <= NHP_SHARDS) __hip_atomic_store(&counter[32 * tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- several parameter sets per launch ---------------------------------------------------------------
// At short windows an evaluation is bound by the scattered gathers of the parent windows, which do
// not depend on the parameters.  nhp_cont_loglik_batch therefore evaluates S = 2 or 4 models of the same kinds
// in ONE pass over the data: S columns of the tables sit in LDS, every parent record is fetched once and used S
// times.  Same arithmetic per model as k_windowed (same item layout, same per-child lane order); the logs of a
// round are taken by all lanes from a small LDS buffer.
#define NHP_MULTI_MAX 8
struct nhp_multi {
    const double *p1[NHP_MULTI_MAX], *p2[NHP_MULTI_MAX], *W[NHP_MULTI_MAX], *A[NHP_MULTI_MAX], *lambda0[NHP_MULTI_MAX], *grid[NHP_MULTI_MAX];
    double *out[NHP_MULTI_MAX];
};

__device__ __forceinline__ double baseline_at_p(int kind, const double *lambda0, const double *x, int grid_n, int c, double t)
{
#pragma clang fp contract(off)
    if (kind == NHP_BASELINE_HOMOGENEOUS) return lambda0[c];
    const double *y = lambda0 + (size_t)c * grid_n;
    int lo = 0, hi = grid_n - 1;
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

template <int IMP, int G, int S>
__global__ __launch_bounds__(NHP_WBLOCK) void k_windowed_multi(nhp_cont_args a, nhp_multi mm, double *__restrict__ partials,
                                                              unsigned int *__restrict__ counter)
{
    constexpr int U = 4, GROUPS = NHP_WBLOCK / G, GW = 64 / G, ROUND = GROUPS * U;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [16] + flag
    double2 *col = reinterpret_cast<double2 *>(smem + 192);         // [S][N]
    double *colw = reinterpret_cast<double *>(col + (size_t)S * a.N);   // [S][N], logit-normal only
    double *rb = colw + (IMP == NHP_IMPULSE_EXPONENTIAL ? 0 : (size_t)S * a.N);   // [ROUND][S] λ of the round's children
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x;
    double integ[S];
#pragma unroll
    for (int m = 0; m < S; ++m) {
        integ[m] = 0.0;
        for (int p = tid; p < N; p += NHP_WBLOCK) {
            const size_t k = (size_t)p + (size_t)c * N;
            double w = mm.W[m][k];
            if (mm.A[m]) w = mm.A[m][k] * w;                         // windowed path: the integral is masked too
            if (IMP == NHP_IMPULSE_EXPONENTIAL) {
                col[(size_t)m * N + p] = make_double2(mm.p1[m][k], w);
            } else {
                col[(size_t)m * N + p] = make_double2(mm.p1[m][k], __builtin_sqrt(mm.p2[m][k]));
                colw[(size_t)m * N + p] = w;
            }
            if (it.first) integ[m] += a.cnt[p] * w;
        }
    }
    __syncthreads();
    const int gid = tid / G, gl = tid % G;
    const int nchild = it.kend - it.kbeg;
    const int slot0 = (gid / GW) * (GW * U) + (gid % GW);
    double acc = 0.0;                                               // Σ log λ of model (tid % S)
    for (int r0 = 0; r0 < nchild; r0 += ROUND) {
        double t[U], s[U][S];
        int j[U], f[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = r0 + slot0 + u * GW;
            valid[u] = kk < nchild;
            const nhp_child ch = a.child_w[it.kbeg + (valid[u] ? kk : r0)];
            t[u] = ch.t;
            j[u] = ch.idx - 1 - gl;
            f[u] = valid[u] ? ch.first : 0x7fffffff;
#pragma unroll
            for (int m = 0; m < S; ++m) s[u][m] = 0.0;
        }
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) more |= j[u] >= f[u];
        nhp_event e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = a.ev[j[u] > 0 ? j[u] : 0];
        while (more) {
            nhp_event en[U];
#pragma unroll
            for (int u = 0; u < U; ++u) en[u] = a.ev[j[u] - G > 0 ? j[u] - G : 0];
            asm volatile("" ::: "memory");          // keeps the prefetch a prefetch (see k_windowed_batch)
            more = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double dt = t[u] - e[u].t;
                const bool in = j[u] >= f[u];
#pragma unroll
                for (int m = 0; m < S; ++m) {
                    const double2 q = col[(size_t)m * N + e[u].node];
                    double term;
                    if (IMP == NHP_IMPULSE_EXPONENTIAL) term = q.y * nhp_pdf_exponential(q.x, dt);
                    else term = colw[(size_t)m * N + e[u].node] * nhp_pdf_logitnormal(q.x, q.y, a.inv_dtmax, dt);
                    s[u][m] += in ? term : 0.0;
                }
                j[u] -= G;
                more |= j[u] >= f[u];
                e[u] = en[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < S; ++m) {
                const double v = group_sum<G>(s[u][m]);
                if (gl == 0) {
                    // rb slot = position of the child inside the round; invalid slots hold 1.0 (log 1 = 0)
                    const int slot = slot0 + u * GW;
                    rb[slot * S + m] = valid[u] ? baseline_at_p(a.baseline_kind, mm.lambda0[m], mm.grid[m], a.grid_n, c, t[u]) + v : 1.0;
                }
            }
        __syncthreads();
        for (int e2 = tid; e2 < ROUND * S; e2 += NHP_WBLOCK) acc += nhp_log(rb[e2]);      // e2 % S == tid % S
        __syncthreads();
    }
    // per-model workgroup sums (lane tid holds model tid % S), then the fused last-block finalize per model
    int *flag = reinterpret_cast<int *>(red + 16);
#pragma unroll
    for (int m = 0; m < S; ++m) {
        const double blk = nhp_block_sum_n<NHP_WBLOCK / 64>((tid % S) == m ? acc : 0.0, red);
        const double blk_int = nhp_block_sum_n<NHP_WBLOCK / 64>(integ[m], red);
        if (tid == 0) {
            __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * m], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * m + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SHARDS - 1) / NHP_SHARDS;
        const unsigned int used = nb < NHP_SHARDS ? nb : NHP_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
#pragma unroll
    for (int m = 0; m < S; ++m) {
        double sl = 0.0, si = 0.0;
        for (unsigned int i = tid; i < gridDim.x; i += NHP_WBLOCK) {
            sl += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            si += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * m + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        nhp_cont_args am = a;
        am.lambda0 = mm.lambda0[m]; am.grid = mm.grid[m];
        double sb = baseline_integral_part(am);
        sl = nhp_block_sum_n<NHP_WBLOCK / 64>(sl, red);
        si = nhp_block_sum_n<NHP_WBLOCK / 64>(si, red);
        sb = nhp_block_sum_n<NHP_WBLOCK / 64>(sb, red);
        if (tid == 0) *mm.out[m] = (0.0 - sb) - si + sl;
    }
    if (tid <= NHP_SHARDS) __hip_atomic_store(&counter[32 * tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int IMP, int S>
static void launch_multi(int G, dim3 grid, size_t lds, hipStream_t st, const nhp_cont_args &a, const nhp_multi &mm,
                         double *partials, unsigned int *counter)
{
#define NHP_MCASE(g)                                                                                              \
    case g:                                                                                                       \
        if (lds > 64 * 1024)                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_windowed_multi<IMP, g, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed_multi<IMP, g, S>), grid, dim3(NHP_WBLOCK), lds, st, a, mm, partials, counter); \
        break;
    switch (G) {
        NHP_MCASE(1) NHP_MCASE(2) NHP_MCASE(4) NHP_MCASE(8) NHP_MCASE(16)
    default:
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)k_windowed_multi<IMP, 32, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_windowed_multi<IMP, 32, S>), grid, dim3(NHP_WBLOCK), lds, st, a, mm, partials, counter);
    }
#undef NHP_MCASE
}

// S models (2 or 4) of identical kinds on one dataset, results into ctx->d_results[slot0 .. slot0+S)
template <int S>
static nhp_status enqueue_multi(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *const *ms, int32_t slot0)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const nhp_cont_model *m0 = ms[0];
    const bool expo = m0->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    const int G = ds->group > 32 ? 32 : ds->group;
    const size_t round = (size_t)(NHP_WBLOCK / G) * 4;
    const size_t lds = 192 + (expo ? 16 : 24) * (size_t)ds->N * S + 8 * round * S;
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items * S));
    nhp_cont_args a = nhp_make_args(ds, m0);
    nhp_multi mm;
    for (int k = 0; k < NHP_MULTI_MAX; ++k) {
        const nhp_cont_model *m = ms[k < S ? k : 0];
        mm.p1[k] = m->d_p1; mm.p2[k] = m->d_p2; mm.W[k] = m->d_W; mm.A[k] = m->has_A ? m->d_A : nullptr;
        mm.lambda0[k] = m->d_lambda0; mm.grid[k] = m->d_grid;
        mm.out[k] = ctx->d_results + slot0 + (k < S ? k : 0);
    }
    dim3 grid((unsigned)ds->n_items);
    if (expo) launch_multi<NHP_IMPULSE_EXPONENTIAL, S>(G, grid, lds, ctx->stream, a, mm, ctx->d_partials, ctx->d_counter);
    else launch_multi<NHP_IMPULSE_LOGITNORMAL, S>(G, grid, lds, ctx->stream, a, mm, ctx->d_partials, ctx->d_counter);
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

// ---- S parameter sets per launch, one lane per (child, parameter set) ---------------------------------------------
// The real callers of "log-likelihood evaluations per second" are batches: the 2P objective calls of a finite-difference
// gradient inside mle! (src/continuous.jl:190), restarts, chain populations.  For S models on one dataset the
// parameter-independent part of an evaluation -- the scattered gathers of the parent windows, the child records --
// is paid once: a workgroup stages column c of all S tables in LDS (S x 16 KB at N = 1024), and lane (k, m) walks the
// window of child k for model m, most recent parent first (the reference's own order, src/continuous.jl:290-298), with
// no cross-lane reduction at all: the S lanes of a child load the same parent record (one request), read their own
// model's {θ, a·w} from LDS, and keep their own sum, their own log.  What is left per evaluation is its S-independent
// share of the gathers plus its own exponentials -- the fp64 VALU, not the fabric, becomes the bound.
// LDS image: model m's column at m·(N+1) double2s: the +1 skews the S models of one child onto different banks.
// PACK: the loaders fetch the 8-byte parent records (k_windowed) and expand them to {t, node} when they park them in the
// window buffer -- once per record, not once per term -- so only the global fetch changes.
template <int IMP, int S, int THREADS, bool PACK = false>
__global__ __launch_bounds__(THREADS) void k_windowed_batch(nhp_cont_args a, nhp_multi mm, double *__restrict__ partials,
                                                           unsigned int *__restrict__ counter)
{
    constexpr int NW = THREADS / 64, CW = 64 / S;                   // waves; children a wave holds at a time
    constexpr int CSTR = S + 1;                                     // records per child row of the window buffer (+1: bank skew)
    static_assert(S == 2 || S == 4 || S == 8, "lane = (child, model) with S a power of two <= 8");
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2][NW][S] wave sums, then the flag
    int *flag = reinterpret_cast<int *>(red + 2 * NW * S);
    double *etab = reinterpret_cast<double *>(smem + 16 * (NW * S + 1));                 // [64] 2^(j/64) for nhp_exp_neg_tab
    nhp_exp_tab_init(etab);
    nhp_event *wbuf = reinterpret_cast<nhp_event *>(smem + 16 * (NW * S + 1) + 512);    // [NW][CW][CSTR] staged parent records
    double2 *col = reinterpret_cast<double2 *>(wbuf + NW * CW * CSTR);                   // [S][N+1]
    const int N = a.N, NP = N + 1, tid = threadIdx.x;
    double *colw = reinterpret_cast<double *>(col + (size_t)S * NP);                     // [S][N+1], logit-normal only
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node;
    const int wv = tid >> 6, ln = tid & 63;
    const int m = ln % S, kc = ln / S;                              // this lane's model, and child inside the wave's group

    NHP_STAMP_AT(0);
    // ---- stage column c of the S models; the node's first item also owns the columns' integral terms.  All loads of a
    // pass over p are issued before any is consumed: the workgroup is alone on its CU (the S columns fill the LDS), so a
    // chain of S dependent global round trips here is time nothing else hides.
    double integ = 0.0;                                             // lane q of every wave: the wave's share for model q
    {
        double part[S];
#pragma unroll
        for (int q = 0; q < S; ++q) part[q] = 0.0;
        for (int p = tid; p < (NHP_SKIP(a, 2) ? 0 : N); p += THREADS) {
            const size_t k = (size_t)p + (size_t)c * N;
            double w[S], a1[S], a2[S];
#pragma unroll
            for (int q = 0; q < S; ++q) {
                w[q] = mm.W[q][k];
                a1[q] = mm.p1[q][k];
                if (IMP != NHP_IMPULSE_EXPONENTIAL) a2[q] = mm.p2[q][k];
                if (mm.A[q]) w[q] *= mm.A[q][k];                     // windowed path: the integral is masked too
            }
            const double cp = it.first ? a.cnt[p] : 0.0;
#pragma unroll
            for (int q = 0; q < S; ++q) {
                if (IMP == NHP_IMPULSE_EXPONENTIAL) {
                    col[(size_t)q * NP + p] = make_double2(a1[q] * -92.33248261689366, w[q] * a1[q]);     // {-θ·64/ln2, a·w·θ}: the term is y·exp2((x·Δt)/64)
                } else {
                    col[(size_t)q * NP + p] = make_double2(a1[q], __builtin_sqrt(a2[q]));
                    colw[(size_t)q * NP + p] = w[q];
                }
                part[q] += cp * w[q];
            }
        }
        // slot N of every column: the "no parent" entry (rate 0, weight 0 -> the pair term is exactly 0) that the padding
        // records of the window buffer point at, so the pair loop has no predicate
        if (tid < S) {
            col[(size_t)tid * NP + N] = make_double2(0.0, IMP == NHP_IMPULSE_EXPONENTIAL ? 0.0 : 1.0);
            if (IMP != NHP_IMPULSE_EXPONENTIAL) colw[(size_t)tid * NP + N] = 0.0;
        }
        if (it.first) {                                             // workgroup-uniform
#pragma unroll
            for (int q = 0; q < S; ++q) {
                const double v = nhp_wave_sum(part[q]);
                if (ln == q) integ = v;
            }
        }
    }
    __syncthreads();
    NHP_STAMP_AT(1);

    const double2 *mycol = col + (size_t)m * NP;
    const double *mycolw = colw + (size_t)m * NP;
    const double *l0 = mm.lambda0[m], *gx = mm.grid[m];
    const int nchild = it.kend - it.kbeg;
    // Σ log λ over this lane's children as ONE logarithm: log Π λ_k = log(Π mant_k) + ln2 · Σ exp_k with λ_k = mant_k · 2^exp_k
    // (v_frexp_mant_f64 / v_frexp_exp_i32_f64, 4 instructions per child instead of the ~45 of a software fp64 log); the
    // running product of mantissas in [0.5, 1) is renormalised after every child group.  A negative λ poisons the product
    // with a NaN, as its logarithm would.
    double prod = 1.0;
    int pexp = 0;
    const bool homog = a.baseline_kind == NHP_BASELINE_HOMOGENEOUS;
    const double base_h = homog ? l0[c] : 0.0;
    nhp_event *myrow = wbuf + ((size_t)wv * CW + kc) * CSTR;        // this child's staged records; as loader: slot m of the row

    // The item's children are sorted by window length; groups of CW consecutive children go to the waves round by round,
    // odd rounds in reverse wave order (serpentine), so every wave gets long and short windows alike: the workgroup -- the
    // only one on its CU, the S columns fill the LDS -- ends when its slowest wave does.
    const int ngroups = (nchild + CW - 1) / CW;
    const int nrounds = NHP_SKIP(a, 1) ? 0 : (ngroups + NW - 1) / NW;
    // group of this wave in round r (-1: none), its child record, its first chunk of parents
    auto group_of = [&](int r) {
        const int g = r * NW + ((r & 1) ? NW - 1 - wv : wv);
        return (r < nrounds && g < ngroups) ? g : -1;
    };
    auto load_child = [&](int g) {
        const int ci = g * CW + kc;
        nhp_child ch = a.child_w[it.kbeg + (g >= 0 ? (ci < nchild ? ci : g * CW) : 0)];
        if (g < 0 || ci >= nchild) ch.first = ch.idx;               // no window: every staged record is padding
        return ch;
    };
    // As a LOADER this lane fetches parent number c0 + m of its child (most recent first): the S lanes of a child read S
    // consecutive records -- one contiguous run, one request -- and park them in the child's row of the window buffer,
    // where all S model-lanes then read them back (same address: a broadcast, no bank conflict).  Parents past the window
    // are stored as {t, node N}: Δt = 0 on the zero-weight entry.
    struct raw { nhp_event e; unsigned long long w; };              // (one of the two is loaded)
    auto fetch = [&](const nhp_child &ch, int c0) {
        const int jj = ch.idx - 1 - (c0 + m);
        raw x;
        if (PACK) x.w = a.ev8[jj > 0 ? jj : 0];
        else x.e = a.ev[jj > 0 ? jj : 0];
        return x;
    };
    // the record as it is parked (expanded where the load is CONSUMED: next to the load it would wait for it)
    auto settle = [&](const raw &x, const nhp_child &ch, int c0) {
        nhp_event e;
        if (PACK) {
            const double d = __hiloint2double((int)(((unsigned)(x.w >> 32) & 0xFFFFu) | 0x43300000u), (int)(unsigned)x.w);
            e.t = __builtin_fma(d - 4503599627370496.0, a.ev8_inv, a.ev8_t0);
            e.node = (int)(x.w >> 48);
            e.pad = 0;
        } else {
            e = x.e;
        }
        if (c0 + m >= ch.idx - ch.first) { e.t = ch.t; e.node = N; }
        e.t = ch.t - e.t;                                           // parked as the DELAY: one subtraction per record, not per (record, set)
        return e;
    };
    // Two dependent global round trips lead into a round (child record, then its parents) and a round is only a few
    // hundred cycles of arithmetic.  Rounds are therefore taken RB at a time: all RB child records are requested, then all
    // RB first chunks of parents, then the arithmetic of the RB rounds runs with nothing left to wait for (register
    // arrays with compile-time indices; a rotating software pipeline would wait at every register hand-over).
    constexpr int RB = 4;
    for (int rb = 0; rb < nrounds; rb += RB) {
        nhp_child chs[RB];
        raw m0s[RB], m1s[RB];
        int kmaxs[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) chs[rr] = load_child(group_of(rb + rr));
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) m0s[rr] = fetch(chs[rr], 0);
        // the second chunk of a round too, where the wave's longest window needs one (wave-uniform; the item is sorted by
        // window length, so it is whole waves that do or do not)
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            kmaxs[rr] = nhp_wave_max_i32(chs[rr].idx - chs[rr].first);
            if (kmaxs[rr] > S) m1s[rr] = fetch(chs[rr], S);
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int g = group_of(rb + rr);
            if (g < 0) continue;                                    // wave-uniform
            const nhp_child ch0 = chs[rr];
            const bool valid = g * CW + kc < nchild;
            const int kmax = kmaxs[rr];                             // longest window of the wave's CW children
            double s = 0.0;
            raw mine = m0s[rr];
            for (int c0 = 0; c0 < kmax; c0 += S) {
                NHP_LDS_SYNC();                                      // the previous chunk's reads precede this overwrite
                myrow[m] = settle(mine, ch0, c0);
                if (c0 == 0) mine = m1s[rr];                         // (already requested above)
                else if (c0 + S < kmax) mine = fetch(ch0, c0 + S);   // third and later chunks: in flight under this chunk's math
                NHP_LDS_SYNC();
                const int nrec = kmax - c0 < S ? kmax - c0 : S;      // wave-uniform
                // records in pairs: the two terms of a pair are independent instruction streams the scheduler interleaves
                // (one term is a chain of ~20 dependent fp64 operations behind two LDS reads; four waves per SIMD alone do
                // not cover it).  The odd record of a last pair is padding at worst: zero weight, no predicate.
#pragma unroll
                for (int r = 0; r < S; r += 2) {
                    if (r < nrec && !NHP_SKIP(a, 8)) {
                        const nhp_event e0 = myrow[r], e1 = myrow[r + 1];
                        const double d0 = e0.t, d1 = e1.t;
                        const double2 q0 = mycol[e0.node], q1 = mycol[e1.node];
                        double t0, t1;
                        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
                            t0 = q0.y * nhp_exp_neg_tab_scaled(q0.x * d0, etab);
                            t1 = q1.y * nhp_exp_neg_tab_scaled(q1.x * d1, etab);
                        } else {
                            t0 = mycolw[e0.node] * nhp_pdf_logitnormal(q0.x, q0.y, a.inv_dtmax, d0);
                            t1 = mycolw[e1.node] * nhp_pdf_logitnormal(q1.x, q1.y, a.inv_dtmax, d1);
                        }
                        s += t0;
                        s += t1;
                    }
                }
            }
            double lam = (homog ? base_h : baseline_at_p(a.baseline_kind, l0, gx, a.grid_n, c, ch0.t)) + s;
            lam = valid ? lam : 1.0;
            const double mt = __builtin_amdgcn_frexp_mant(lam);
            prod *= lam < 0.0 ? __builtin_nan("") : mt;
            pexp += __builtin_amdgcn_frexp_exp(lam) + __builtin_amdgcn_frexp_exp(prod);
            prod = __builtin_amdgcn_frexp_mant(prod);
        }
    }
    NHP_STAMP_AT(2);
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;   // (log 0 = -Inf, log NaN = NaN: as the per-child logs)
    if (prod == 0.0) acc = -__builtin_inf();
    // ---- per-model sums: lanes with equal (lane % S) inside the wave, then the waves in a fixed order
    for (int off = S; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (ln < S) { red[wv * S + ln] = acc; red[(NW + wv) * S + ln] = integ; }
    __syncthreads();
    NHP_STAMP_AT(3);
    if (tid < S) {
        double blk = 0.0, blk_int = 0.0;
        for (int w = 0; w < NW; ++w) { blk += red[w * S + tid]; blk_int += red[(NW + w) * S + tid]; }
        __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * tid], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * tid + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                                // every storing lane has drained before the ticket
    if (tid == 0) {
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SHARDS - 1) / NHP_SHARDS;
        const unsigned int used = nb < NHP_SHARDS ? nb : NHP_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    NHP_STAMP_AT(4);
    if (!*flag) return;
    // the last workgroup adds all partials of model (tid % S) in a fixed order; THREADS / S lanes per model
    {
        constexpr int CPR = THREADS / S;
        const int kslot = tid / S, mt = tid % S;
        const double *l0t = mm.lambda0[mt], *gxt = mm.grid[mt];
        double sl = 0.0, si = 0.0;
        for (unsigned int i = kslot; i < gridDim.x; i += CPR) {
            sl += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * mt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            si += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * mt + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // baseline integral of model mt, spread over the CPR lanes of the model
        double sb = 0.0;
        for (int cc = a.col_begin + kslot; cc < a.col_end; cc += CPR) {
            if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) {
                sb += l0t[cc] * a.duration;
            } else {
                const double *y = l0t + (size_t)cc * a.grid_n;
                double I = 0.0;
                for (int i = 0; i + 1 < a.grid_n; ++i) I += 0.5 * (y[i] + y[i + 1]) * (gxt[i + 1] - gxt[i]);
                sb += I;
            }
        }
        for (int off = S; off < 64; off <<= 1) {
            sl += __shfl_xor(sl, off, 64); si += __shfl_xor(si, off, 64); sb += __shfl_xor(sb, off, 64);
        }
        __syncthreads();
        double *r3 = reinterpret_cast<double *>(wbuf);              // window buffer + columns are dead: [3][NW][S]
        if (ln < S) { r3[wv * S + ln] = sl; r3[(NW + wv) * S + ln] = si; r3[(2 * NW + wv) * S + ln] = sb; }
        __syncthreads();
        if (tid < S) {
            double tl = 0.0, ti = 0.0, tb = 0.0;
            for (int w = 0; w < NW; ++w) { tl += r3[w * S + tid]; ti += r3[(NW + w) * S + tid]; tb += r3[(2 * NW + w) * S + tid]; }
            *mm.out[tid] = (0.0 - tb) - ti + tl;
        }
    }
    if (tid <= NHP_SHARDS) __hip_atomic_store(&counter[32 * tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup size of k_windowed_batch: 8 sets fill the LDS of a CU with one workgroup of 16 waves; 4 (or 2) sets leave room
// for two workgroups of 8 waves, whose latency phases (launch, staging, record fetches, reduction tail) overlap
#define NHP_BATCH_THREADS(S) ((S) >= 8 ? 1024 : 512)
template <int S>
static size_t batch_lds(const nhp_cont_dataset *ds, bool expo, int threads)
{
    // wave sums + flag | window buffer [waves][64/S children][S+1 records] | S columns (the finalizing workgroup reuses the
    // window buffer for its 3 x waves x S sums: 24·waves·S <= 16·waves·(64/S)·(S+1) bytes for every S)
    return 16 * ((size_t)(threads / 64) * S + 1) + 512 + 16 * (size_t)(threads / 64) * (64 / S) * (S + 1) + (expo ? 16 : 24) * (size_t)(ds->N + 1) * S;

}

// S models (2, 4 or 8) of identical kinds on one dataset, results into ctx->d_results[slot0 .. slot0+S)
template <int S>
static nhp_status enqueue_batch(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *const *ms, int32_t slot0)
{
    constexpr int THREADS = NHP_BATCH_THREADS(S);
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const nhp_cont_model *m0 = ms[0];
    const bool expo = m0->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    const size_t lds = batch_lds<S>(ds, expo, THREADS);
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items * S));
    nhp_cont_args a = nhp_make_args(ds, m0);
    nhp_multi mm;
    for (int k = 0; k < NHP_MULTI_MAX; ++k) {
        const nhp_cont_model *m = ms[k < S ? k : 0];
        mm.p1[k] = m->d_p1; mm.p2[k] = m->d_p2; mm.W[k] = m->d_W; mm.A[k] = m->has_A ? m->d_A : nullptr;
        mm.lambda0[k] = m->d_lambda0; mm.grid[k] = m->d_grid;
        mm.out[k] = ctx->d_results + slot0 + (k < S ? k : 0);
    }
    dim3 grid((unsigned)ds->n_items);
    if (expo && a.ev8) {
        if (lds > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_windowed_batch<NHP_IMPULSE_EXPONENTIAL, S, THREADS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_windowed_batch<NHP_IMPULSE_EXPONENTIAL, S, THREADS, true>), grid, dim3(THREADS), lds, ctx->stream, a, mm, ctx->d_partials, ctx->d_counter);
    } else if (expo) {
        if (lds > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_windowed_batch<NHP_IMPULSE_EXPONENTIAL, S, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_windowed_batch<NHP_IMPULSE_EXPONENTIAL, S, THREADS>), grid, dim3(THREADS), lds, ctx->stream, a, mm, ctx->d_partials, ctx->d_counter);
    } else {
        if (lds > 64 * 1024) NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_windowed_batch<NHP_IMPULSE_LOGITNORMAL, S, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_windowed_batch<NHP_IMPULSE_LOGITNORMAL, S, THREADS>), grid, dim3(THREADS), lds, ctx->stream, a, mm, ctx->d_partials, ctx->d_counter);
    }
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

// ll = -Σ_c ∫λ0_c - Σ_blocks integ + Σ_blocks loglam   (src/continuous.jl:216-221,237)
// Baseline integral: λ .* duration (src/baselines.jl:98-102) or the trapezoid rule over the
// grid, which ignores `duration` (src/baselines.jl:336, src/utils/interpolation.jl:40-48).
__global__ __launch_bounds__(NHP_BLOCK) void k_finalize(nhp_cont_args a, const double *__restrict__ partials,
                                                        int n_partials, double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    double sl = 0.0, si = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += NHP_BLOCK) {
        sl += partials[2 * (size_t)i];
        si += partials[2 * (size_t)i + 1];
    }
    sb = baseline_integral_part(a);
    sl = nhp_block_sum(sl, red);
    si = nhp_block_sum(si, red);
    sb = nhp_block_sum(sb, red);
    if (threadIdx.x == 0) *out = (0.0 - sb) - si + sl;
}

template <int IMP>
static void launch_group(int G, dim3 grid, size_t lds, hipStream_t st, const nhp_cont_args &a, int mask,
                         double *partials, double *lambda_out, unsigned int *counter, double *out)
{
#define NHP_CASE(g, u)                                                                                        \
    case g:                                                                                                   \
        if (lds > 64 * 1024)                                                                                  \
            (void)hipFuncSetAttribute((const void *)k_windowed<IMP, g, u>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed<IMP, g, u>), grid, dim3(NHP_WBLOCK), lds, st, a, mask, partials,       \
                           lambda_out, counter, out);                                                         \
        break;
    // short and middle windows, exponential impulses, no λ output: the 8-byte parent records (the fetches bound these launches;
    // K = 64: 133.9 -> 124.1 us)
    if (IMP == NHP_IMPULSE_EXPONENTIAL && a.ev8 && !lambda_out && G <= (getenv("NHP_PACK_G") ? atoi(getenv("NHP_PACK_G")) : 16)) {   // (wider groups are VALU-bound: the decode costs more than the bytes save: K = 512 570 -> 611 us)
        constexpr int EI = NHP_IMPULSE_EXPONENTIAL;
#define NHP_PCASE(g)                                                                                          \
    case g:                                                                                                   \
        if (lds > 64 * 1024)                                                                                  \
            (void)hipFuncSetAttribute((const void *)k_windowed<EI, g, NHP_U_SMALL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed<EI, g, NHP_U_SMALL, true>), grid, dim3(NHP_WBLOCK), lds, st, a, mask, partials, lambda_out, counter, out); \
        return;
        // half the bytes per record move the best width down: 4 lanes per child where the 16-byte records want 8
        // (N = 1024, M = 1e6, K = 8: G = 8 37.0 us, G = 4 32.4, G = 2 42.7; tools/kbench.py under NHP_GROUP)
        const int Gp = G == 8 && !getenv("NHP_GROUP") ? 4 : G;
        switch (Gp) { NHP_PCASE(1) NHP_PCASE(2) NHP_PCASE(4) NHP_PCASE(8) NHP_PCASE(16) NHP_PCASE(32) default: break; }
#undef NHP_PCASE
    }
    switch (G) {
        NHP_CASE(1, NHP_U_SMALL) NHP_CASE(2, NHP_U_SMALL) NHP_CASE(4, NHP_U_SMALL) NHP_CASE(8, NHP_U_SMALL) NHP_CASE(16, NHP_U_MID) NHP_CASE(32, NHP_U_MID)
    default:
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)k_windowed<IMP, 64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_windowed<IMP, 64, 1>), grid, dim3(NHP_WBLOCK), lds, st, a, mask, partials, lambda_out, counter, out);
    }
#undef NHP_CASE
}

static nhp_status run_windowed(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m,
                               double *d_out, double *d_lambda, const nhp_child *child_w = nullptr, int group = 0,
                               int mask_integral = 1)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? 16 : 24;
    const size_t lds = 192 + per * (size_t)ds->N + 512;            // + the 64-entry exp table
    if (lds > 160 * 1024) { nhp_set_error(ctx, "n_nodes = %d exceeds the 160 KiB LDS column budget", ds->N); return NHP_ENOTIMPL; }
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items));
    nhp_cont_args a = nhp_make_args(ds, m);
    if (child_w) a.child_w = child_w;                     // same children, other window starts (recursive path)
    const int G = group ? group : ds->group;
    dim3 grid((unsigned)ds->n_items);
    // the dataset's own short windows: through the cached pair list (k_windowed_pairs)
    const bool plist_off = getenv("NHP_PLIST") && atoi(getenv("NHP_PLIST")) == 0;       // (read per call: the tests switch it)
    const bool expo_p = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    // ... one lane per child over the child slices (cont_slices.hip) when there is no λ output: 6-byte records
    if (!child_w && !d_lambda && d_out && expo_p && !plist_off) {
        bool launched = false;
        NHP_TRY(nhp_launch_windowed_slices(ctx, ds, m, mask_integral, d_out, &launched));
        if (launched) return NHP_OK;
    }
    if (!child_w && (d_out || d_lambda) && m->impulse_kind == NHP_IMPULSE_LOGITNORMAL && !plist_off) {     // ... and the logit-normal twin (λ out as well)
        bool launched = false;
        NHP_TRY(nhp_launch_windowed_slices_ln(ctx, ds, m, mask_integral, d_out, d_lambda, &launched));
        if (launched) return NHP_OK;
    }
    if (!child_w && ds->d_poff && !plist_off && G <= 16 && (expo_p || nhp_ensure_pair_cache(ctx, ds, &a) == NHP_OK)) {
        if (expo_p && !ds->d_plist) {                              // first use: build the list (data only)
            nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
            if (hipMalloc((void **)&mds->d_plist, 8 * (size_t)std::max<int64_t>(ds->pairs, 1)) != hipSuccess) {
                mds->d_plist = nullptr;
                (void)hipGetLastError();
            } else {
                hipLaunchKernelGGL(k_pairs_build, dim3((unsigned)((ds->M + 255) / 256)), dim3(256), 0, ctx->stream, a, mds->d_plist);
                NHP_HIP(ctx, hipGetLastError());
                a.plist = mds->d_plist;
            }
        }
        const size_t lds2 = 320 + (expo_p ? 16 : 24) * (size_t)ds->N + 512 + 4 * ((size_t)ds->max_item + 1) + 16;
        if ((expo_p ? ds->d_plist != nullptr : ds->d_plq != nullptr) && lds2 <= 160 * 1024) {
            // lanes per child, children per group in flight, workgroup size: measured at N = 1024, M = 1e6, K = 8
            // (tools/dbg/pairsweep.sh); NHP_PAIRS_CFG = "G,U,BLOCK" overrides
            int Gp = G >= 16 ? 8 : G == 8 ? 4 : G, Up = 1, Bp = 512;                // (with a round requested ahead, one child per group in flight: 24.6 us; two: 26.3)
            if (const char *cfg = getenv("NHP_PAIRS_CFG")) sscanf(cfg, "%d,%d,%d", &Gp, &Up, &Bp);
            bool ok = false;
#define NHP_LCASE(g, u, b)                                                                                    \
    if (!ok && Gp == g && Up == u && Bp == b) {                                                               \
        ok = true;                                                                                            \
        if (expo_p) {                                                                                         \
            if (lds2 > 64 * 1024)                                                                             \
                (void)hipFuncSetAttribute((const void *)k_windowed_pairs<NHP_IMPULSE_EXPONENTIAL, g, u, b>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
            hipLaunchKernelGGL((k_windowed_pairs<NHP_IMPULSE_EXPONENTIAL, g, u, b>), grid, dim3(b), lds2, ctx->stream, a, mask_integral, ds->max_item, \
                               ctx->d_partials, d_lambda, ctx->d_counter, d_out);                                       \
        } else {                                                                                              \
            if (lds2 > 64 * 1024)                                                                             \
                (void)hipFuncSetAttribute((const void *)k_windowed_pairs<NHP_IMPULSE_LOGITNORMAL, g, u, b>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
            hipLaunchKernelGGL((k_windowed_pairs<NHP_IMPULSE_LOGITNORMAL, g, u, b>), grid, dim3(b), lds2, ctx->stream, a, mask_integral, ds->max_item, \
                               ctx->d_partials, d_lambda, ctx->d_counter, d_out);                                       \
        }                                                                                                     \
    }
#define NHP_LROW(g) NHP_LCASE(g, 1, 256) NHP_LCASE(g, 2, 256) NHP_LCASE(g, 4, 256) NHP_LCASE(g, 1, 512) NHP_LCASE(g, 2, 512) NHP_LCASE(g, 4, 512) \
                    NHP_LCASE(g, 1, 1024) NHP_LCASE(g, 2, 1024)
            NHP_LROW(1) NHP_LROW(2) NHP_LROW(4) NHP_LROW(8)
            if (!ok) { Gp = 4; Up = 1; Bp = 512; NHP_LCASE(4, 1, 512) }
#undef NHP_LROW
#undef NHP_LCASE
            NHP_HIP(ctx, hipGetLastError());
            return NHP_OK;
        }
    }
    if (m->impulse_kind == NHP_IMPULSE_EXPONENTIAL)
        launch_group<NHP_IMPULSE_EXPONENTIAL>(G, grid, lds, ctx->stream, a, mask_integral, ctx->d_partials, d_lambda, ctx->d_counter, d_out);
    else
        launch_group<NHP_IMPULSE_LOGITNORMAL>(G, grid, lds, ctx->stream, a, mask_integral, ctx->d_partials, d_lambda, ctx->d_counter, d_out);
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

nhp_status nhp_launch_finalize(nhp_ctx *ctx, const nhp_cont_args &a, int n_partials, double *d_out)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(NHP_BLOCK), 0, ctx->stream, a, ctx->d_partials, n_partials, d_out);
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

nhp_status nhp_launch_windowed(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out)
{
    return run_windowed(ctx, ds, m, d_out, nullptr);
}

nhp_status nhp_launch_windowed_as(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child *child_w,
                                  int group, int mask_integral, double *d_out)
{
    return run_windowed(ctx, ds, m, d_out, nullptr, child_w, group, mask_integral);
}

nhp_status nhp_launch_event_intensity_as(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child *child_w,
                                         int group, int mask_integral, double *d_lambda)
{
    return run_windowed(ctx, ds, m, nullptr, d_lambda, child_w, group, mask_integral);
}

nhp_status nhp_launch_event_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_lambda)
{
    return run_windowed(ctx, ds, m, nullptr, d_lambda);
}

// ---- C ABI ------------------------------------------------------------------------------

static nhp_status enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags, int32_t slot)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (slot < 0 || slot >= NHP_MAX_SLOTS) return NHP_EINVAL;
    if ((flags & NHP_LL_RECURSIVE) && m->impulse_kind == NHP_IMPULSE_EXPONENTIAL)
        return nhp_launch_recursive_flags(ctx, ds, m, flags, ctx->d_results + slot);
    return nhp_launch_windowed(ctx, ds, m, ctx->d_results + slot);
}

extern "C" nhp_status nhp_cont_loglik_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                              const nhp_cont_model *m, int32_t flags, int32_t slot)
{
    return enqueue(ctx, ds, m, flags, slot);
}

extern "C" nhp_status nhp_cont_loglik(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m,
                                      int32_t flags, double *ll)
{
    if (!ll) return NHP_EINVAL;
    NHP_TRY(enqueue(ctx, ds, m, flags, 0));
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}

// nb evaluations (finite-difference sweeps, chain populations) back to back, one synchronisation
// models that one k_windowed_multi launch can take together: same kinds and shapes, all windowed
static bool multi_compatible(const nhp_cont_dataset *ds, const nhp_cont_model *x, const nhp_cont_model *y, size_t lds_limit, int S)
{
    if (!x || !y || x->ctx != y->ctx || x->N != y->N || x->impulse_kind != y->impulse_kind || x->baseline_kind != y->baseline_kind ||
        x->grid_n != y->grid_n || x->has_A != y->has_A || x->dt_max != y->dt_max) return false;
    const size_t lds = 192 + (x->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? 16 : 24) * (size_t)ds->N * S + 8 * 256 * 4 * (size_t)S;
    return lds <= lds_limit;
}

extern "C" nhp_status nhp_cont_loglik_batch(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                            const nhp_cont_model *const *models, int32_t nb, int32_t flags, double *ll)
{
    if (!models || !ll || nb < 0) return NHP_EINVAL;
    // windowed evaluations at short windows are bound by gathers that do not depend on the parameters: take the models
    // four (or two) at a time through one pass over the data; everything else goes one launch per model
    static const int fuse = getenv("NHP_BATCH_FUSE") ? atoi(getenv("NHP_BATCH_FUSE")) : 8;      // largest group: 0/1 off, 2, 4, 8
    static const bool batch_kernel = getenv("NHP_BATCH_KERNEL") ? atoi(getenv("NHP_BATCH_KERNEL")) != 0 : true;   // 0: the older k_windowed_multi
    static const bool two_lanes = getenv("NHP_BATCH_LANES") ? atoi(getenv("NHP_BATCH_LANES")) >= 2 : true;
    const double kbar = ds && ds->M > 0 ? (double)ds->pairs / (double)ds->M : 0.0;
    // Windowed launches alternate between the context's two lanes (stream + partial sums + tickets each): they are
    // independent -- different result slots, read-only data and models -- so the second lane forks from the main stream
    // (everything enqueued before this call, e.g. parameter uploads, is visible to it) and joins it before the results
    // are fetched.  The lane is switched by swapping the context's fields around a launch, so the launch code is shared.
    struct lane_guard {
        nhp_ctx *c; bool on = false;
        void flip() { std::swap(c->stream, c->stream2); std::swap(c->d_partials, c->d_partials2); std::swap(c->partials_cap, c->partials2_cap);
                      std::swap(c->d_counter, c->d_counter2); on = !on; }
        ~lane_guard() { if (on) flip(); }
    } lane{ctx};
    bool forked = false;
    int launches = 0;
    for (int32_t done = 0; done < nb; done += NHP_MAX_SLOTS) {
        const int32_t n = nb - done < NHP_MAX_SLOTS ? nb - done : NHP_MAX_SLOTS;
        int32_t k = 0;
        while (k < n) {
            const nhp_cont_model *const *ms = models + done + k;
            const bool windowed = ms[0] && !((flags & NHP_LL_RECURSIVE) && ms[0]->impulse_kind == NHP_IMPULSE_EXPONENTIAL);
            int take = 1;
            bool lane_kernel = false;                      // k_windowed_batch (one lane per (child, model)) vs k_windowed_multi
            if (fuse >= 2 && windowed && kbar <= 96.0 && ds && ds->N >= 1) {
                NHP_TRY(nhp_check_pair(ctx, ds, ms[0]));
                const bool expo0 = ms[0]->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
                auto all_compatible = [&](int S) {
                    if (k + S > n) return false;
                    for (int q = 1; q < S; ++q)
                        if (!multi_compatible(ds, ms[0], ms[q], (size_t)1 << 30, S)) return false;
                    return true;
                };
                if (batch_kernel) {
                    const size_t cap = 160 * 1024;
                    if (fuse >= 8 && all_compatible(8) && batch_lds<8>(ds, expo0, NHP_BATCH_THREADS(8)) <= cap) take = 8;
                    else if (fuse >= 4 && all_compatible(4) && batch_lds<4>(ds, expo0, NHP_BATCH_THREADS(4)) <= cap) take = 4;
                    else if (all_compatible(2) && batch_lds<2>(ds, expo0, NHP_BATCH_THREADS(2)) <= cap) take = 2;
                    lane_kernel = take > 1;
                } else if (kbar <= 48.0) {
                    if (fuse >= 4 && k + 4 <= n && multi_compatible(ds, ms[0], ms[1], 80 * 1024, 4) && multi_compatible(ds, ms[0], ms[2], 80 * 1024, 4) &&
                        multi_compatible(ds, ms[0], ms[3], 80 * 1024, 4)) take = 4;
                    else if (k + 2 <= n && multi_compatible(ds, ms[0], ms[1], 64 * 1024, 2)) take = 2;
                }
            }
            const bool second = two_lanes && windowed && ctx && ctx->stream2 && (launches & 1);
            if (second) {
                if (!forked) {
                    NHP_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                    NHP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                    forked = true;
                }
                lane.flip();
            }
            for (int q = 1; q < take; ++q) NHP_TRY(nhp_check_pair(ctx, ds, ms[q]));
            // exponential models on a dataset with child slices: four (or two) at a time through ONE pass over the slices
            // (cont_slices.hip: every pair record fetched and decoded once for all of them)
            if (windowed && take >= 2 && ds->d_sl_row) {
                const int S = take >= 4 ? 4 : 2;
                bool launched = false;
                NHP_TRY(nhp_launch_slices_batch(ctx, ds, ms, S, k, &launched));
                if (launched) {
                    if (second) lane.flip();
                    ++launches;
                    k += S;
                    continue;
                }
            }
            if (take == 8) NHP_TRY(enqueue_batch<8>(ctx, ds, ms, k));
            else if (take == 4 && lane_kernel) NHP_TRY(enqueue_batch<4>(ctx, ds, ms, k));
            else if (take == 2 && lane_kernel) NHP_TRY(enqueue_batch<2>(ctx, ds, ms, k));
            else if (take == 4) NHP_TRY(enqueue_multi<4>(ctx, ds, ms, k));
            else if (take == 2) NHP_TRY(enqueue_multi<2>(ctx, ds, ms, k));
            else NHP_TRY(enqueue(ctx, ds, ms[0], flags, k));
            if (second) lane.flip();
            if (windowed) ++launches;
            k += take;
        }
        if (forked) {                                       // the main stream continues after the second lane's launches
            NHP_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            NHP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            forked = false;
        }
        NHP_TRY(nhp_ctx_fetch(ctx, 0, n, ll + done));
    }
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_event_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds,
                                               const nhp_cont_model *m, double *lambda)
{
    if (!lambda) return NHP_EINVAL;
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    NHP_WHOLE_DATASET(ctx, ds, "event_intensity");
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, sizeof(double) * (size_t)(ds->M > 0 ? ds->M : 1)));
    NHP_TRY(nhp_launch_event_intensity(ctx, ds, m, (double *)ctx->d_scratch));
    NHP_HIP(ctx, hipMemcpyAsync(lambda, ctx->d_scratch, sizeof(double) * (size_t)ds->M, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NHP_OK;
}
