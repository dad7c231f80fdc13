// Windowed exponential log-likelihood over CHILD SLICES: one lane per child event.
//   λ_c(t_i) = λ0_c(t_i) + Σ_{j<i, t_j > t_i-Δtmax} A[n_j,c] W[n_j,c] θ[n_j,c] exp(-θ[n_j,c] (t_i - t_j))
// (reference: loglikelihood src/continuous.jl:210-239,360-389; total_intensity :286-300 -- the backwards walk from event
//  i-1 that adds one impulse per parent is exactly what a lane does here, most recent parent first).
//
// Mapping to CDNA4
//  * A workgroup owns an item (a run of children of ONE node c): column c of the tables is staged once in LDS as
//    {-θ·unit·64/ln2, a·w·θ} per parent node, entry N = {0, 0} for the padding records.
//  * A wavefront owns a slice = 64 consecutive children of the item (children are sorted by window length, so the 64
//    windows are nearly equal).  Which (parent, child) pairs exist and their delays are data, not parameters; the dataset
//    keeps them slice by slice, ROW r = the r-th most recent parent of each of the 64 children: one wave-wide load fetches a
//    row (256 + 128 contiguous bytes in the two planes of the 6-byte records), every lane keeps its own running sum, and
//    there is no cross-lane reduction, no per-pair predicate and no per-child offset arithmetic at all.
//  * Per evaluation the kernel streams 6 bytes per (padded) pair + 16 bytes per parameter pair: 67 MB at N = 1024, M = 1e6,
//    mean window 8 (the 8-byte list of k_windowed_pairs: 85 MB), and issues ~25 VALU instructions per pair instead of ~44.
//  * Σ log λ as one logarithm per lane of a running mantissa product; block sums and the fused last-workgroup reduction as
//    in k_windowed.
#include <algorithm>
#include <vector>

#include "nhp_internal.h"
#include "nhp_math.h"
#include "nhp_rng.h"

struct nhp_slices {               // kernel-side view of nhp_cont_dataset::d_sl_*
    const uint32_t *row;          // [n_slices + 1]
    const int32_t *item0;         // [n_items + 1]
    const uint32_t *lo;           // [(rows + 16) * 64]
    const uint16_t *hi;
    int32_t nsh;                  // 16 - node bits: hi = node << nsh | delay >> 32
    uint32_t dmask;               // (1 << nsh) - 1
    int32_t dbits;                // bits of the delay: 32 + nsh
};

#define NHP_SL_SHARDS 64

#ifdef NHP_STAMP      // diagnostic build only (tools/dbg/slstamps.py): s_memrealtime (100 MHz, one clock for all XCDs) at the phase boundaries of wave 0 of every workgroup
__device__ unsigned long long g_sl_stamps[8 * 4096];
#define NHP_SL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sl_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int nhp_debug_stamps_slices(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sl_stamps), sizeof(unsigned long long) * (size_t)n);
}
#else
#define NHP_SL_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double sl_baseline(const nhp_cont_args &a, int c, double t)
{
#pragma clang fp contract(off)
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return a.lambda0[c];
    const double *x = a.grid;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    int lo = 0, hi = a.grid_n - 1;
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

// -∫λ0_c: λ·duration (src/baselines.jl:98-102) or the trapezoid rule over the grid, which ignores `duration`
// (src/baselines.jl:336); spread over the threads of the workgroup (their block sum adds it up)
__device__ __forceinline__ double sl_baseline_integral_col(const nhp_cont_args &a, int c)
{
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return threadIdx.x == 0 ? a.lambda0[c] * a.duration : 0.0;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    double I = 0.0;
    for (int i = threadIdx.x; i + 1 < a.grid_n; i += blockDim.x) I += 0.5 * (y[i] + y[i + 1]) * (a.grid[i + 1] - a.grid[i]);
    return I;
}

// ∂ log λ_i / ∂(grid intensities of node c) for an event at time t: g = 1/λ_i spread over the two grid neighbours with the
// interpolation weights (src/utils/interpolation.jl:26-35); as grad_lgcp_scatter of cont_grad.hip
__device__ __forceinline__ void sl_lgcp_scatter(const nhp_cont_args &a, int c, double t, double g, double *grad)
{
    const double *x = a.grid;
    const int G = a.grid_n;
    double *gc = grad + (size_t)c * G;
    int lo = 0, hi = G - 1;
    if (!(t < x[hi])) { atomicAdd(&gc[hi], g); return; }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    const double w = x[lo + 1] - x[lo];
    atomicAdd(&gc[lo], g * (x[lo + 1] - t) / w);
    atomicAdd(&gc[lo + 1], g * (t - x[lo]) / w);
}

// Fills the planes: wave = slice, lane = child, row r = the child's r-th most recent parent (src/continuous.jl:290-298 walks
// the window in this order).  Delays are rounded to 2^-dbits of Δtmax and kept inside [1, 2^dbits - 1] (a tie Δt = 0 becomes
// one unit: 7e-12·Δtmax at N = 1024); rows past a child's window hold {node N, delay 0}.
__global__ __launch_bounds__(256) void k_slices_build(nhp_cont_args a, nhp_slices sl, uint32_t *__restrict__ lo, uint16_t *__restrict__ hi)
{
    const nhp_item it = a.items[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    const double two_d = __builtin_ldexp(1.0, sl.dbits);
    const double scale = a.inv_dtmax * two_d, qmax = two_d - 1.0;
    for (int j = w; j < ns; j += 4) {
        const uint32_t row0 = sl.row[s0 + j];
        const int K = (int)(sl.row[s0 + j + 1] - row0);
        const int kk = 64 * j + lane;
        nhp_child ch;
        ch.t = 0.0; ch.first = 0; ch.idx = 0;
        if (kk < nchild) ch = a.child_w[it.kbeg + kk];
        const int len = ch.idx - ch.first;
        for (int r = 0; r < K; ++r) {
            uint32_t l = 0, h = (uint32_t)a.N << sl.nsh;
            if (r < len) {
                const nhp_event e = a.ev[ch.idx - 1 - r];
                double q = __builtin_rint((ch.t - e.t) * scale);
                q = q < 1.0 ? 1.0 : (q > qmax ? qmax : q);
                const uint64_t qi = (uint64_t)q;
                l = (uint32_t)qi;
                h = ((uint32_t)e.node << sl.nsh) | (uint32_t)(qi >> 32);
            }
            const size_t o = ((size_t)row0 + (size_t)r) * 64 + (size_t)lane;
            lo[o] = l;
            hi[o] = (uint16_t)h;
        }
    }
}

struct nhp_pslices {              // kernel-side view of the parent slices (nhp_cont_dataset::d_ps_*) + where the gradient goes
    const uint32_t *row;          // [n_items * spi + 1]
    const uint16_t *perm;         // [n_items * spi * 64]
    const uint32_t *lo;
    const uint16_t *hi;
    int32_t spi;                  // parent slices per item
    int32_t psh;                  // 16 - slot bits: hi = slot << psh | delay >> 32
    uint32_t dmask;
    int32_t dbits;
    int32_t max_item;             // the padding records' slot (its 1/λ is 0)
    int32_t direct;               // every item is its node's only one, whole dataset, flat baseline: the gradient is stored, not added
    double *grad;                 // [P] params! order
};

// pairs of every item by parent node: cnt[item * N + p]
__global__ __launch_bounds__(256) void k_ps_count(nhp_cont_args a, uint32_t *__restrict__ cnt)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const nhp_item it = a.items[blockIdx.x];
    for (int p = threadIdx.x; p < a.N; p += 256) hist[p] = 0;
    __syncthreads();
    for (int k = it.kbeg + threadIdx.x; k < it.kend; k += 256) {
        const nhp_child ch = a.child_w[k];
        for (int j = ch.idx - 1; j >= ch.first; --j) atomicAdd(&hist[a.ev[j].node], 1u);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < a.N; p += 256) cnt[(size_t)blockIdx.x * a.N + p] = hist[p];
}

// the item's parent nodes by pair count (most first, ties by node): lane order of the parent slices, rows of every slice
__global__ __launch_bounds__(256) void k_ps_rank(int N, int spi, const uint32_t *__restrict__ cnt, uint16_t *__restrict__ perm,
                                                 uint16_t *__restrict__ rankof, uint32_t *__restrict__ rows)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *c = reinterpret_cast<uint32_t *>(smem);
    for (int p = threadIdx.x; p < N; p += 256) c[p] = cnt[(size_t)blockIdx.x * N + p];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * spi * 64;
    for (int p = threadIdx.x; p < N; p += 256) {
        const uint32_t mine = c[p];
        int rank = 0;
        for (int q = 0; q < N; ++q) rank += (c[q] > mine || (c[q] == mine && q < p)) ? 1 : 0;
        perm[base + rank] = (uint16_t)p;
        rankof[(size_t)blockIdx.x * N + p] = (uint16_t)rank;
        if ((rank & 63) == 0) rows[(size_t)blockIdx.x * spi + (rank >> 6)] = mine;
    }
    for (int q = N + threadIdx.x; q < spi * 64; q += 256) perm[base + q] = 0xFFFFu;
}

// every pair into its parent's lane (position by an LDS ticket: any order), then the rest of the lane padded
__global__ __launch_bounds__(256) void k_ps_fill(nhp_cont_args a, nhp_pslices ps, const uint32_t *__restrict__ cnt, const uint16_t *__restrict__ rankof,
                                                 uint32_t *__restrict__ lo, uint16_t *__restrict__ hi)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *cursor = reinterpret_cast<uint32_t *>(smem);
    const nhp_item it = a.items[blockIdx.x];
    const int N = a.N;
    for (int p = threadIdx.x; p < N; p += 256) cursor[p] = 0;
    __syncthreads();
    const double two_d = __builtin_ldexp(1.0, ps.dbits);
    const double scale = a.inv_dtmax * two_d, qmax = two_d - 1.0;
    const uint32_t *row = ps.row + (size_t)blockIdx.x * ps.spi;
    for (int k = it.kbeg + threadIdx.x; k < it.kend; k += 256) {
        const nhp_child ch = a.child_w[k];
        const uint32_t slot = (uint32_t)(k - it.kbeg);
        for (int j = ch.idx - 1; j >= ch.first; --j) {
            const nhp_event e = a.ev[j];
            const uint32_t pos = atomicAdd(&cursor[e.node], 1u);
            const int rk = rankof[(size_t)blockIdx.x * N + e.node];
            double q = __builtin_rint((ch.t - e.t) * scale);
            q = q < 1.0 ? 1.0 : (q > qmax ? qmax : q);
            const uint64_t qi = (uint64_t)q;
            const size_t o = ((size_t)row[rk >> 6] + pos) * 64 + (size_t)(rk & 63);
            lo[o] = (uint32_t)qi;
            hi[o] = (uint16_t)((slot << ps.psh) | (uint32_t)(qi >> 32));
        }
    }
    // lanes shorter than their slice, and the lanes behind the last node
    for (int rk = threadIdx.x; rk < ps.spi * 64; rk += 256) {
        const int p = ps.perm[(size_t)blockIdx.x * ps.spi * 64 + rk];
        const uint32_t n = p == 0xFFFF ? 0u : cnt[(size_t)blockIdx.x * N + p];
        const uint32_t r0 = row[rk >> 6], R = row[(rk >> 6) + 1] - r0;
        for (uint32_t i = n; i < R; ++i) {
            const size_t o = ((size_t)r0 + i) * 64 + (size_t)(rk & 63);
            lo[o] = 0;
            hi[o] = (uint16_t)((uint32_t)ps.max_item << ps.psh);
        }
    }
}

// every lane's records sorted by (slot, delay): the list is the same whatever order the tickets were drawn in
__global__ __launch_bounds__(256) void k_ps_sort(int N, nhp_pslices ps, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ lo,
                                                 uint16_t *__restrict__ hi)
{
    const uint32_t *row = ps.row + (size_t)blockIdx.x * ps.spi;
    for (int rk = threadIdx.x; rk < ps.spi * 64; rk += 256) {
        const int p = ps.perm[(size_t)blockIdx.x * ps.spi * 64 + rk];
        if (p == 0xFFFF) continue;
        const int n = (int)cnt[(size_t)blockIdx.x * N + p];
        const size_t o0 = (size_t)row[rk >> 6] * 64 + (size_t)(rk & 63);
        for (int i = 1; i < n; ++i) {
            const uint32_t l = lo[o0 + (size_t)i * 64], h = hi[o0 + (size_t)i * 64];
            const uint64_t key = ((uint64_t)h << 32) | l;
            int j = i - 1;
            while (j >= 0) {
                const uint32_t lj = lo[o0 + (size_t)j * 64], hj = hi[o0 + (size_t)j * 64];
                if ((((uint64_t)hj << 32) | lj) <= key) break;
                lo[o0 + (size_t)(j + 1) * 64] = lj;
                hi[o0 + (size_t)(j + 1) * 64] = (uint16_t)hj;
                --j;
            }
            lo[o0 + (size_t)(j + 1) * 64] = l;
            hi[o0 + (size_t)(j + 1) * 64] = (uint16_t)h;
        }
    }
}

// C rows of a slice are requested at a time into one of two register sets: the next set is in flight while this one is
// summed, across slice boundaries too (the first rows of a wave's next slice are requested under the last rows of this one,
// and the very first set before the column is staged: its addresses need the slice table only).
// GRAD: the analytic gradient in the same launch (the objective of mle!, src/continuous.jl:144-198, differentiated by hand;
// formulas at the top of cont_grad.hip).  Phase A is the log-likelihood itself and leaves g_k = 1/λ_k of the item's children
// in LDS; phase B walks the item's PARENT slices -- lane = parent node p, its θ[p,c], W[p,c] in registers, row r = the node's
// r-th pair as {child slot, delay} -- and every lane sums Σ g·e^{-θΔ} and Σ g·(1-θΔ)·e^{-θΔ} over its own pairs: no atomics,
// no cross-lane step, the list streamed once.  An item that is its node's only one stores the column's gradient (the
// parameter-independent terms included), others add to what k_grad_init left.
template <int BLOCK, int C, bool FLAT, bool GRAD>
__global__ __launch_bounds__(BLOCK) void k_windowed_slices(nhp_cont_args a, nhp_slices sl, nhp_pslices ps, int mask_integral,
                                                            double *__restrict__ partials, unsigned int *__restrict__ counter,
                                                            double *__restrict__ out)
{
    constexpr int NW = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2 * NW <= 32] + flag at [32]
    double2 *col = reinterpret_cast<double2 *>(smem + 320);         // [N + 1] {-θ·unit·64/ln2, a·w·θ}; [N] = {0, 0}
    double *etab = reinterpret_cast<double *>(col + a.N + 1);       // [64] 2^(j/64)
    double *ginv = etab + 64;                                       // GRAD: [max_item + 1] 1/λ of the item's children; [max_item] = 0
    const int tid = threadIdx.x, lane = tid & 63;
    const double tab_v = nhp_exp2_64[lane];                         // requested first, parked in LDS with the column (one wait)

    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    NHP_SL_STAMP(0);

    // Rows r .. r+C-1 of the slice that starts at row `row0`: a wave-uniform base and compile-time row offsets.  Rows past the
    // slice's last are simply the next slice's (or the 16 rows of padding behind the list): loaded, never summed.
    struct chunk { uint32_t lo[C], hi[C]; };
    auto request = [&](chunk &q, const uint32_t *plo, const uint16_t *phi, const uint32_t row0, const int r) {
        const size_t o = ((size_t)row0 + (size_t)r) * 64;
        const uint32_t *pl = plo + o;
        const uint16_t *ph = phi + o;
#pragma unroll
        for (int u = 0; u < C; ++u) {
            q.lo[u] = pl[u * 64 + lane];
            q.hi[u] = ph[u * 64 + lane];
        }
    };
    // Column c of the tables: every load of (up to) UN passes over the parent nodes is requested before any is used -- one
    // round trip instead of one per pass -- and the first rows of the wave's first slice are requested right behind them
    // (their addresses need the slice table only), so they arrive while the column is being written to LDS.
    constexpr int UN = BLOCK >= 512 ? 2 : 4;
    const double unit = __builtin_ldexp(a.dt_max, -sl.dbits);      // Δtmax · 2^-dbits
    int j = w;
    uint32_t row0 = 0;
    int K = 0;
    chunk qa, qb;
    double integ = 0.0;
    for (int p0 = 0; p0 < N; p0 += UN * BLOCK) {
        double w_[UN], th_[UN], a_[UN], cnt_[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = p0 + u * BLOCK + tid;
            const size_t k = (size_t)(p < N ? p : 0) + (size_t)c * N;
            w_[u] = a.W[k];
            th_[u] = a.p1[k];
            a_[u] = a.A ? a.A[k] : 1.0;
            cnt_[u] = it.first ? a.cnt[p < N ? p : 0] : 0.0;
        }
        if (p0 == 0) {
            if (j < ns) { row0 = sl.row[s0 + j]; K = (int)(sl.row[s0 + j + 1] - row0); }
            asm volatile("" ::: "memory");
            request(qa, sl.lo, sl.hi, row0, 0);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = p0 + u * BLOCK + tid;
            if (p < N) {
                double wv = w_[u], wint = wv;
                if (a.A) {
                    wv = a_[u] * wv;
                    if (mask_integral) wint = wv;
                }
                col[p] = make_double2(-((th_[u] * unit) * 92.33248261689366), wv * th_[u]);   // term = (a·w·θ)·exp(-(θ·unit)·q), the rate times 64/ln 2
                integ += cnt_[u] * wint;
            }
        }
    }
    if (tid < 64) etab[tid] = tab_v;
    if (tid == 0) col[N] = make_double2(0.0, 0.0);
    if (GRAD && tid == 0) ginv[ps.max_item] = 0.0;
    if (out && it.first) integ += sl_baseline_integral_col(a, c);
    __syncthreads();
    NHP_SL_STAMP(1);

    const double lam0 = FLAT ? a.lambda0[c] : 0.0;
    const uint32_t dmask = sl.dmask;
    const int nsh = sl.nsh;
    // the high delay bits under the exponent of 2^52 (one v_bfi_b32): the double 2^52 + delay, minus 2^52
    auto delay = [&](const uint32_t lo, const uint32_t h, const uint32_t mask) {
        uint32_t hw;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hw) : "s"(mask), "v"(h), "v"(0x43300000u));
        return __hiloint2double((int)hw, (int)lo) - 4503599627370496.0;
    };
    auto term = [&](const uint32_t lo, const uint32_t h) {
        const double v = delay(lo, h, dmask);
        const double2 cw = col[h >> nsh];
        return cw.y * nhp_exp_neg_tab_scaled(cw.x * v, etab);
    };
    // A request that lies wholly inside the slice is summed as C independent straight-line chains (their LDS round trips --
    // column entry, then the 2^(j/64) entry -- overlap); the slice's last request goes row by row behind wave-uniform
    // branches.  The sum itself is sequential in both: most recent parent first, the reference's order.
    auto sum = [&](const chunk &q, const int r, const int K, double s) {
        if (r + C <= K) {
            double t[C];
#pragma unroll
            for (int u = 0; u < C; ++u) t[u] = term(q.lo[u], q.hi[u]);
#pragma unroll
            for (int u = 0; u < C; ++u) s += t[u];
        } else {
#pragma unroll
            for (int u = 0; u < C; ++u)
                if (r + u < K) s += term(q.lo[u], q.hi[u]);
        }
        return s;
    };
    double prod = 1.0, gsum = 0.0;
    int pexp = 0;
    while (j < ns) {
        const int jn = j + NW;
        uint32_t row0n = 0;
        int Kn = 0;
        if (jn < ns) { row0n = sl.row[s0 + jn]; Kn = (int)(sl.row[s0 + jn + 1] - row0n); }
        double s = 0.0;
        if (K <= 0) request(qa, sl.lo, sl.hi, row0n, 0);
        for (int r0 = 0; r0 < K; r0 += 2 * C) {
            request(qb, sl.lo, sl.hi, row0, r0 + C);
            asm volatile("" ::: "memory");
            s = sum(qa, r0, K, s);
            const bool more = r0 + 2 * C < K;
            request(qa, sl.lo, sl.hi, more ? row0 : row0n, more ? r0 + 2 * C : 0);
            asm volatile("" ::: "memory");
            s = sum(qb, r0 + C, K, s);
        }
        const int kk = 64 * j + lane;
        if (kk < nchild) {
            const double tk = FLAT ? 0.0 : a.child_w[it.kbeg + kk].t;
            const double lam = (FLAT ? lam0 : sl_baseline(a, c, tk)) + s;
            prod *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
            pexp += __builtin_amdgcn_frexp_exp(lam);
            if (GRAD) {
                const double g = 1.0 / lam;
                ginv[kk] = g;
                if (FLAT) gsum += g;
                else sl_lgcp_scatter(a, c, tk, g, ps.grad);
            }
        }
        pexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
        j = jn; row0 = row0n; K = Kn;
    }
    NHP_SL_STAMP(2);
    if (GRAD) {
        // ---- phase B: the item's pairs by parent node ----
        const uint32_t *prow = ps.row + (size_t)blockIdx.x * ps.spi;
        const uint16_t *pperm = ps.perm + (size_t)blockIdx.x * ps.spi * 64;
        const double punit = __builtin_ldexp(a.dt_max, -ps.dbits);
        const uint32_t pmask = ps.dmask;
        const int psh = ps.psh;
        const size_t nbase = a.baseline_kind == NHP_BASELINE_HOMOGENEOUS ? (size_t)N : (size_t)N * (size_t)a.grid_n;
        const size_t NN = (size_t)N * (size_t)N;
        // a lane's node and its parameters, requested a slice ahead
        struct lanep { int p; double th, wv, av, cn; };
        auto lane_params = [&](const int jj) {
            lanep q;
            q.p = jj < ps.spi ? (int)pperm[jj * 64 + lane] : 0xFFFF;
            const size_t k = (size_t)(q.p == 0xFFFF ? 0 : q.p) + (size_t)c * N;
            q.th = a.p1[k];
            q.wv = a.W[k];
            q.av = a.A ? a.A[k] : 1.0;
            q.cn = ps.direct ? a.cnt[q.p == 0xFFFF ? 0 : q.p] : 0.0;
            return q;
        };
        int jp = w;
        uint32_t prow0 = 0;
        int Kp = 0;
        if (jp < ps.spi) { prow0 = prow[jp]; Kp = (int)(prow[jp + 1] - prow0); }
        lanep cur = lane_params(jp);
        request(qa, ps.lo, ps.hi, prow0, 0);
        __syncthreads();                                            // every wave's g_k are in LDS
        auto gterm = [&](const uint32_t lo, const uint32_t h, const double xs, double &accH, double &acc1) {
            const double v = delay(lo, h, pmask);
            const double g = ginv[h >> psh];
            const double t = xs * v;
            const double ge = g * nhp_exp_neg_tab_scaled(t, etab);
            accH += ge;                                             // Σ g·e^{-θΔ}
            acc1 = __builtin_fma(ge, __builtin_fma(t, 1.0830424696249145e-02, 1.0), acc1);   // Σ g·(1 - θΔ)·e^{-θΔ}; θΔ = -t·ln2/64
        };
        auto gsum_chunk = [&](const chunk &q, const int r, const int K, const double xs, double &accH, double &acc1) {
            if (r + C <= K) {
#pragma unroll
                for (int u = 0; u < C; ++u) gterm(q.lo[u], q.hi[u], xs, accH, acc1);
            } else {
#pragma unroll
                for (int u = 0; u < C; ++u)
                    if (r + u < K) gterm(q.lo[u], q.hi[u], xs, accH, acc1);
            }
        };
        while (jp < ps.spi) {
            const int jn = jp + NW;
            uint32_t row0n = 0;
            int Kn = 0;
            if (jn < ps.spi) { row0n = prow[jn]; Kn = (int)(prow[jn + 1] - row0n); }
            const lanep nxt = lane_params(jn);
            const double xs = -((cur.th * punit) * 92.33248261689366);
            double accH = 0.0, acc1 = 0.0;
            if (Kp <= 0) request(qa, ps.lo, ps.hi, row0n, 0);
            for (int r0 = 0; r0 < Kp; r0 += 2 * C) {
                request(qb, ps.lo, ps.hi, prow0, r0 + C);
                asm volatile("" ::: "memory");
                gsum_chunk(qa, r0, Kp, xs, accH, acc1);
                const bool more = r0 + 2 * C < Kp;
                request(qa, ps.lo, ps.hi, more ? prow0 : row0n, more ? r0 + 2 * C : 0);
                asm volatile("" ::: "memory");
                gsum_chunk(qb, r0 + C, Kp, xs, accH, acc1);
            }
            if (cur.p != 0xFFFF) {
                const size_t k = (size_t)cur.p + (size_t)c * N;
                const double gW = cur.av * (cur.th * accH);         // ∂/∂W[p,c]: a Σ g·θ e^{-θΔ}
                const double gT = (cur.av * cur.wv) * acc1;         // ∂/∂θ[p,c]: a·w Σ g·(1 - θΔ) e^{-θΔ}
                if (ps.direct) {
                    const double mk = (a.A && mask_integral) ? cur.av : 1.0;
                    ps.grad[nbase + NN + k] = gW - cur.cn * mk;
                    ps.grad[nbase + k] = gT;
                } else {
                    if (gW != 0.0) atomicAdd(&ps.grad[nbase + NN + k], gW);
                    if (gT != 0.0) atomicAdd(&ps.grad[nbase + k], gT);
                }
            }
            jp = jn; prow0 = row0n; Kp = Kn; cur = nxt;
        }
    }
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;
    if (prod == 0.0) acc = -__builtin_inf();
    double blk = acc, blk_int = integ;
    nhp_block_sum2_n<NW>(blk, blk_int, red);
    if (GRAD && FLAT) {
        const double gs = nhp_block_sum_n<NW>(gsum, red);
        if (tid == 0) {
            if (ps.direct) ps.grad[c] = gs - a.duration;
            else if (gs != 0.0) atomicAdd(&ps.grad[c], gs);
        }
        __syncthreads();
    }
    NHP_SL_STAMP(3);
    if (!out) {
        if (tid == 0) {
            partials[2 * (size_t)blockIdx.x] = blk;
            partials[2 * (size_t)blockIdx.x + 1] = blk_int;
        }
        return;
    }
    // fused second stage (as k_windowed): write-through partials, one sharded ticket; the workgroup that draws the last
    // ticket adds all partials in a fixed order and leaves the tickets at 0 for the next launch
    int *flag = reinterpret_cast<int *>(red + 32);
    if (tid == 0) {
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SL_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SL_SHARDS - 1) / NHP_SL_SHARDS;
        const unsigned int used = nb < NHP_SL_SHARDS ? nb : NHP_SL_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    NHP_SL_STAMP(4);
    if (!*flag) return;
    double sl_ = 0.0, si = 0.0;
    for (unsigned int i = tid; i < gridDim.x; i += BLOCK) {
        sl_ += __hip_atomic_load(&partials[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si += __hip_atomic_load(&partials[2 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nhp_block_sum2_n<NW>(sl_, si, red);
    if (tid == 0) *out = (0.0 - si) + sl_;
    for (int i = tid; i <= NHP_SL_SHARDS; i += BLOCK)               // (a one-wave workgroup has 64 threads for the 65 words)
        __hip_atomic_store(&counter[32 * i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- S parameter sets per pass --------------------------------------------------------------------------
// The callers of "log-likelihood evaluations per second" come in batches (the 2P objective calls of a finite-difference
// gradient inside the reference's mle!, src/continuous.jl:190; restarts :200; chain populations).  For S models on one dataset
// a slice's rows are fetched and decoded ONCE; a lane (= child) keeps S running sums, one per model, each term from that
// model's column in LDS (S planes of N + 1 entries).  What a pass then costs is its exponentials: ~18 instructions per pair
// and model + 4 per pair.
#define NHP_SETS_MAX 4
struct nhp_sets {
    const double *p1[NHP_SETS_MAX], *W[NHP_SETS_MAX], *A[NHP_SETS_MAX], *lambda0[NHP_SETS_MAX];
    double *out[NHP_SETS_MAX];
};

template <int BLOCK, int C, int S>
__global__ __launch_bounds__(BLOCK) void k_slices_batch(nhp_cont_args a, nhp_slices sl, nhp_sets st, double *__restrict__ partials,
                                                         unsigned int *__restrict__ counter)
{
    constexpr int NW = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2 * NW <= 32] + flag at [32]
    double2 *col = reinterpret_cast<double2 *>(smem + 320);         // [S][N + 1]
    double *etab = reinterpret_cast<double *>(col + (size_t)S * (a.N + 1));
    const int tid = threadIdx.x, lane = tid & 63;
    const double tab_v = nhp_exp2_64[lane];
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;

    struct chunk { uint32_t lo[C], hi[C]; };
    auto request = [&](chunk &q, const uint32_t row0, const int r) {
        const size_t o = ((size_t)row0 + (size_t)r) * 64;
        const uint32_t *pl = sl.lo + o;
        const uint16_t *ph = sl.hi + o;
#pragma unroll
        for (int u = 0; u < C; ++u) {
            q.lo[u] = pl[u * 64 + lane];
            q.hi[u] = ph[u * 64 + lane];
        }
    };
    int j = w;
    uint32_t row0 = 0;
    int K = 0;
    if (j < ns) { row0 = sl.row[s0 + j]; K = (int)(sl.row[s0 + j + 1] - row0); }
    chunk qa, qb;
    request(qa, row0, 0);

    const double unit = __builtin_ldexp(a.dt_max, -sl.dbits);
    double integ[S];
#pragma unroll
    for (int m = 0; m < S; ++m) {
        integ[m] = 0.0;
        for (int p = tid; p < N; p += BLOCK) {
            const size_t k = (size_t)p + (size_t)c * N;
            double wv = st.W[m][k];
            if (st.A[m]) wv = st.A[m][k] * wv;                      // windowed route: the integral is masked too
            const double th = st.p1[m][k];
            col[(size_t)m * (N + 1) + p] = make_double2(-((th * unit) * 92.33248261689366), wv * th);
            if (it.first) integ[m] += a.cnt[p] * wv;
        }
        if (tid == 0) col[(size_t)m * (N + 1) + N] = make_double2(0.0, 0.0);
        if (it.first && tid == 0) integ[m] += st.lambda0[m][c] * a.duration;
    }
    if (tid < 64) etab[tid] = tab_v;
    __syncthreads();

    const uint32_t dmask = sl.dmask;
    const int nsh = sl.nsh;
    const uint32_t plane = (uint32_t)(N + 1) * 16u;                 // bytes between the models' columns
    auto terms = [&](const uint32_t lo, const uint32_t h, double (&acc)[S]) {
        uint32_t hw;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hw) : "s"(dmask), "v"(h), "v"(0x43300000u));
        const double v = __hiloint2double((int)hw, (int)lo) - 4503599627370496.0;
        const unsigned char *base = reinterpret_cast<const unsigned char *>(col) + ((h >> nsh) << 4);
#pragma unroll
        for (int m = 0; m < S; ++m) {
            const double2 cw = *reinterpret_cast<const double2 *>(base + m * plane);
            acc[m] = __builtin_fma(cw.y, nhp_exp_neg_tab_scaled(cw.x * v, etab), acc[m]);
        }
    };
    auto sum = [&](const chunk &q, const int r, const int K, double (&acc)[S]) {
        if (r + C <= K) {
#pragma unroll
            for (int u = 0; u < C; ++u) terms(q.lo[u], q.hi[u], acc);
        } else {
#pragma unroll
            for (int u = 0; u < C; ++u)
                if (r + u < K) terms(q.lo[u], q.hi[u], acc);
        }
    };
    double lam0[S], prod[S];
    int pexp[S];
#pragma unroll
    for (int m = 0; m < S; ++m) { lam0[m] = st.lambda0[m][c]; prod[m] = 1.0; pexp[m] = 0; }
    while (j < ns) {
        const int jn = j + NW;
        uint32_t row0n = 0;
        int Kn = 0;
        if (jn < ns) { row0n = sl.row[s0 + jn]; Kn = (int)(sl.row[s0 + jn + 1] - row0n); }
        double acc[S];
#pragma unroll
        for (int m = 0; m < S; ++m) acc[m] = 0.0;
        if (K <= 0) request(qa, row0n, 0);
        for (int r0 = 0; r0 < K; r0 += 2 * C) {
            request(qb, row0, r0 + C);
            asm volatile("" ::: "memory");
            sum(qa, r0, K, acc);
            const bool more = r0 + 2 * C < K;
            request(qa, more ? row0 : row0n, more ? r0 + 2 * C : 0);
            asm volatile("" ::: "memory");
            sum(qb, r0 + C, K, acc);
        }
        const int kk = 64 * j + lane;
        if (kk < nchild) {
#pragma unroll
            for (int m = 0; m < S; ++m) {
                const double lam = lam0[m] + acc[m];
                prod[m] *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
                pexp[m] += __builtin_amdgcn_frexp_exp(lam);
            }
        }
#pragma unroll
        for (int m = 0; m < S; ++m) {
            pexp[m] += __builtin_amdgcn_frexp_exp(prod[m]);
            prod[m] = __builtin_amdgcn_frexp_mant(prod[m]);
        }
        j = jn; row0 = row0n; K = Kn;
    }
    int *flag = reinterpret_cast<int *>(red + 32);
#pragma unroll
    for (int m = 0; m < S; ++m) {
        double blk = nhp_log(prod[m]) + (double)pexp[m] * 6.93147180559945286e-01;
        if (prod[m] == 0.0) blk = -__builtin_inf();
        double blk_int = integ[m];
        nhp_block_sum2_n<NW>(blk, blk_int, red);
        if (tid == 0) {
            __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * m], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&partials[(2 * (size_t)blockIdx.x) * S + 2 * m + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SL_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SL_SHARDS - 1) / NHP_SL_SHARDS;
        const unsigned int used = nb < NHP_SL_SHARDS ? nb : NHP_SL_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
#pragma unroll
    for (int m = 0; m < S; ++m) {
        double sl_ = 0.0, si = 0.0;
        for (unsigned int i = tid; i < gridDim.x; i += BLOCK) {
            sl_ += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            si += __hip_atomic_load(&partials[(2 * (size_t)i) * S + 2 * m + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        nhp_block_sum2_n<NW>(sl_, si, red);
        if (tid == 0) *st.out[m] = (0.0 - si) + sl_;
        __syncthreads();
    }
    for (int i = tid; i <= NHP_SL_SHARDS; i += BLOCK)
        __hip_atomic_store(&counter[32 * i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- parent sampler over the child slices, logit-normal impulses (config 3's sweep) -------------------------------------
// resample_parent (src/parents.jl:25-46): weights [A·W·ħ(t_i - t_{i-1}), ..., λ0] most recent parent first, their sequential
// sum, one uniform, the first index whose running sum of quotients w_k / s exceeds it.  A child is one lane, as in k_sampler --
// the order of every addition and division is the contract -- but its records arrive row by row from the slice planes (one
// coalesced load per row and wave instead of 64 cache lines per instruction) and the first CACHE weights wait in LDS, row by
// lane, for the scan.  Padding records weigh exactly 0 and are added like the others: x + 0.0 = x, so a child's sum is the
// sequential sum of ITS weights, bit for bit; the scan stops a lane at its own last parent.
__global__ __launch_bounds__(256) void k_slices_build_lq(nhp_cont_args a, nhp_slices sl, double *__restrict__ L, double *__restrict__ Q)
{
#pragma clang fp contract(off)
    const nhp_item it = a.items[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    for (int j = w; j < ns; j += 4) {
        const uint32_t row0 = sl.row[s0 + j];
        const int K = (int)(sl.row[s0 + j + 1] - row0);
        const int kk = 64 * j + lane;
        nhp_child ch;
        ch.t = 0.0; ch.first = 0; ch.idx = 0;
        if (kk < nchild) ch = a.child_w[it.kbeg + kk];
        const int len = ch.idx - ch.first;
        for (int r = 0; r < K; ++r) {
            double2 d = make_double2(0.0, 0.0);
            if (r < len) d = nhp_logitnormal_data(a.inv_dtmax, ch.t - a.ev[ch.idx - 1 - r].t);
            const size_t o = ((size_t)row0 + (size_t)r) * 64 + (size_t)lane;
            L[o] = d.x;
            Q[o] = d.y;
        }
    }
}

// Exponential impulses: the sampler's weights are a·w·θ·e^{-θΔt} of the EXACT Δt = t_i - t_j (the slices' 37-bit delay would
// change the low bits of a weight and, once in a long while, the index the scan stops at: the sampler's contract is bit
// equality with the sequential reference).  One more plane, the delays as doubles, made once per dataset like L and Q.
__global__ __launch_bounds__(256) void k_slices_build_d(nhp_cont_args a, nhp_slices sl, double *__restrict__ D)
{
    const nhp_item it = a.items[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    for (int j = w; j < ns; j += 4) {
        const uint32_t row0 = sl.row[s0 + j];
        const int K = (int)(sl.row[s0 + j + 1] - row0);
        const int kk = 64 * j + lane;
        nhp_child ch;
        ch.t = 0.0; ch.first = 0; ch.idx = 0;
        if (kk < nchild) ch = a.child_w[it.kbeg + kk];
        const int len = ch.idx - ch.first;
        for (int r = 0; r < K; ++r)
            D[((size_t)row0 + (size_t)r) * 64 + (size_t)lane] = r < len ? ch.t - a.ev[ch.idx - 1 - r].t : 0.0;
    }
}

template <int BLOCK, int CACHE, int IMP>
__global__ __launch_bounds__(BLOCK) void k_sampler_slices(nhp_cont_args a, nhp_slices sl, const double *__restrict__ L, const double *__restrict__ Q,
                                                           const double *__restrict__ u, uint64_t seed, uint64_t step,
                                                           int64_t *__restrict__ parents, int64_t *__restrict__ pnodes,
                                                           int32_t *__restrict__ pn_b, double *__restrict__ dt_b, int *__restrict__ err)
{
#pragma clang fp contract(off)
    constexpr int NW = BLOCK / 64, C = 2;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *col = reinterpret_cast<double2 *>(smem);               // [N + 1] {μ, √τ}; [N] = {0, 0}
    double *colw = reinterpret_cast<double *>(col + a.N + 1);       // [N + 1] a·w; [N] = 0
    double *cache = colw + a.N + 1;                                 // [NW][CACHE][64] the first weights of the slice's children
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    for (int p = tid; p < N; p += BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        double wv = a.W[k];
        if (a.A) wv = a.A[k] * wv;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) col[p] = make_double2(a.p1[k], wv);          // {rate, a·w}: k_sampler<0>'s column
        else col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
        colw[p] = wv;
    }
    if (tid == 0) { col[N] = make_double2(0.0, 0.0); colw[N] = 0.0; }
    __syncthreads();
    double *wc = cache + (size_t)w * CACHE * 64 + lane;
    const int nsh = sl.nsh;
    struct chunk { uint32_t hi[C]; double l[C], q[C]; };
    // (exponential: L is the plane of exact delays, Q is not read)
    auto request = [&](chunk &qq, const uint32_t row0, const int r) {
        const size_t o = ((size_t)row0 + (size_t)r) * 64;
#pragma unroll
        for (int x = 0; x < C; ++x) {
            qq.hi[x] = sl.hi[o + x * 64 + lane];
            qq.l[x] = L[o + x * 64 + lane];
            qq.q[x] = IMP == NHP_IMPULSE_EXPONENTIAL ? 0.0 : Q[o + x * 64 + lane];
        }
    };
    auto weight = [&](const uint32_t h, const double l, const double q) {
        const int p = (int)(h >> nsh);
        const double2 cq = col[p];
        if (IMP == NHP_IMPULSE_EXPONENTIAL) return cq.y * nhp_pdf_exponential(cq.x, l);   // the operations of samp_weight<0>
        return colw[p] * nhp_pdf_logitnormal_cached(cq.x, cq.y, make_double2(l, q));
    };
    for (int j = w; j < ns; j += NW) {
        const uint32_t row0 = sl.row[s0 + j];
        const int K = (int)(sl.row[s0 + j + 1] - row0);
        const int kk = 64 * j + lane;
        const bool valid = kk < nchild;
        nhp_child ch;
        ch.t = 0.0; ch.first = 0; ch.idx = 0;
        if (valid) ch = a.child_w[it.kbeg + kk];
        const int i = ch.idx, nreal = ch.idx - ch.first, n = nreal + 1;
        const double t = ch.t;
        const double base = valid ? sl_baseline(a, c, t) : 1.0;
        // ---- the sum, most recent parent first (0 + w_0 = w_0; the padding rows add 0), the baseline last
        double v = 0.0;
        chunk qa, qb;
        request(qa, row0, 0);
        for (int r0 = 0; r0 < K; r0 += 2 * C) {
            request(qb, row0, r0 + C);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int x = 0; x < C; ++x)
                if (r0 + x < K) {
                    const double wk = weight(qa.hi[x], qa.l[x], qa.q[x]);
                    if (r0 + x < CACHE) wc[(size_t)(r0 + x) * 64] = wk;
                    v = v + wk;
                }
            request(qa, row0, r0 + 2 * C);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int x = 0; x < C; ++x)
                if (r0 + C + x < K) {
                    const double wk = weight(qb.hi[x], qb.l[x], qb.q[x]);
                    if (r0 + C + x < CACHE) wc[(size_t)(r0 + C + x) * 64] = wk;
                    v = v + wk;
                }
        }
        const double s = v + base;
        const bool live = valid && i > 0;                           // index == 1 -> (0, 0): src/parents.jl:26-28
        if (live && (!(s > 0.0) || !(s < __builtin_inf()))) *err = 1;
        const double draw = live ? (u ? u[i] : nhp_philox_uniform(seed, step, (uint64_t)i)) : 0.0;
        // ---- the scan: cp_k = cp_{k-1} + w_k / s while cp_k <= u and k < n - 1 (the lanes of the wave step together; a lane
        //      that has stopped stays stopped).  Weight k of a child: its k-th parent, or the baseline at k = n - 1.
        auto wk_at = [&](const int k) {
            if (k >= nreal) return base;
            if (k < CACHE) return wc[(size_t)k * 64];
            const size_t o = ((size_t)row0 + (size_t)k) * 64 + lane;
            return weight(sl.hi[o], L[o], IMP == NHP_IMPULSE_EXPONENTIAL ? 0.0 : Q[o]);
        };
        int kk2 = 0;
        double cp = live ? wk_at(0) / s : 0.0;
        bool go = live && cp <= draw && kk2 < n - 1;
        while (__ballot(go)) {
            if (go) {
                ++kk2;
                cp = cp + wk_at(kk2) / s;
                go = cp <= draw && kk2 < n - 1;
            }
        }
        if (valid) {
            const int parent = (live && kk2 < n - 1) ? i - 1 - kk2 : -1;
            const int pnode = parent >= 0 ? a.nodes[parent] : -1;
            if (parents) parents[i] = (int64_t)parent + 1;          // 1-based event index, 0 = baseline
            if (pnodes) pnodes[i] = (int64_t)pnode + 1;
            const int kb = a.wpos[it.kbeg + kk];
            pn_b[kb] = pnode;
            dt_b[kb] = parent >= 0 ? t - a.times[parent] : 0.0;
        }
    }
}

// ---- logit-normal impulses over the child slices: log-likelihood -------------------------------------------------------------
// The same walk as k_windowed_slices -- one lane per child, rows of 64 records, most recent parent first -- with the record's
// node from the `hi` plane and the data half of its pdf, {logit(x), 1/(x(1-x))}, from the planes the parent sampler keeps
// (sl_L, sl_Q: k_slices_build_lq), the column {μ, √τ} and a·w in LDS: a term is a·w · nhp_pdf_logitnormal_cached, the pair
// cache's arithmetic.  A separate kernel so that the exponential one's code is not touched.
template <int BLOCK, int C, bool FLAT>
__global__ __launch_bounds__(BLOCK) void k_windowed_slices_ln(nhp_cont_args a, nhp_slices sl, const double *__restrict__ L, const double *__restrict__ Q,
                                                               int mask_integral, double *__restrict__ partials, unsigned int *__restrict__ counter,
                                                               double *__restrict__ out, double *__restrict__ lambda_out)
{
    constexpr int NW = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2 * NW <= 32] + flag at [32]
    double2 *col = reinterpret_cast<double2 *>(smem + 320);         // [N + 1] {μ, √τ}; [N] = {0, 0}
    double *colw = reinterpret_cast<double *>(col + a.N + 1);       // [N + 1] a·w; [N] = 0
    const int tid = threadIdx.x, lane = tid & 63;
    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchild = it.kend - it.kbeg;
    const int s0 = sl.item0[blockIdx.x], ns = sl.item0[blockIdx.x + 1] - s0;
    struct chunk { uint32_t hi[C]; double l[C], q[C]; };
    auto request = [&](chunk &qq, const uint32_t row0, const int r) {
        const size_t o = ((size_t)row0 + (size_t)r) * 64;
#pragma unroll
        for (int u = 0; u < C; ++u) {
            qq.hi[u] = sl.hi[o + u * 64 + lane];
            qq.l[u] = L[o + u * 64 + lane];
            qq.q[u] = Q[o + u * 64 + lane];
        }
    };
    int j = w;
    uint32_t row0 = 0;
    int K = 0;
    chunk qa, qb;
    double integ = 0.0;
    for (int p0 = 0; p0 < N; p0 += BLOCK) {
        const int p = p0 + tid;
        const size_t k = (size_t)(p < N ? p : 0) + (size_t)c * N;
        const double w_ = a.W[k], mu = a.p1[k], tau = a.p2[k], a_ = a.A ? a.A[k] : 1.0, cnt_ = it.first ? a.cnt[p < N ? p : 0] : 0.0;
        if (p0 == 0) {
            if (j < ns) { row0 = sl.row[s0 + j]; K = (int)(sl.row[s0 + j + 1] - row0); }
            asm volatile("" ::: "memory");
            request(qa, row0, 0);
            asm volatile("" ::: "memory");
        }
        if (p < N) {
            double wv = w_, wint = w_;
            if (a.A) { wv = a_ * wv; if (mask_integral) wint = wv; }
            col[p] = make_double2(mu, __builtin_sqrt(tau));
            colw[p] = wv;
            integ += cnt_ * wint;
        }
    }
    if (tid == 0) { col[N] = make_double2(0.0, 0.0); colw[N] = 0.0; }
    if (out && it.first) integ += sl_baseline_integral_col(a, c);
    __syncthreads();
    const double lam0 = FLAT ? a.lambda0[c] : 0.0;
    const int nsh = sl.nsh;
    auto term = [&](const uint32_t h, const double l, const double q) {
        const int p = (int)(h >> nsh);
        const double2 cq = col[p];
        return colw[p] * nhp_pdf_logitnormal_cached(cq.x, cq.y, make_double2(l, q));
    };
    auto sum = [&](const chunk &qq, const int r, const int K, double s) {
        if (r + C <= K) {
            double t[C];
#pragma unroll
            for (int u = 0; u < C; ++u) t[u] = term(qq.hi[u], qq.l[u], qq.q[u]);
#pragma unroll
            for (int u = 0; u < C; ++u) s += t[u];
        } else {
#pragma unroll
            for (int u = 0; u < C; ++u)
                if (r + u < K) s += term(qq.hi[u], qq.l[u], qq.q[u]);
        }
        return s;
    };
    double prod = 1.0;
    int pexp = 0;
    while (j < ns) {
        const int jn = j + NW;
        uint32_t row0n = 0;
        int Kn = 0;
        if (jn < ns) { row0n = sl.row[s0 + jn]; Kn = (int)(sl.row[s0 + jn + 1] - row0n); }
        double s = 0.0;
        if (K <= 0) request(qa, row0n, 0);
        for (int r0 = 0; r0 < K; r0 += 2 * C) {
            request(qb, row0, r0 + C);
            asm volatile("" ::: "memory");
            s = sum(qa, r0, K, s);
            const bool more = r0 + 2 * C < K;
            request(qa, more ? row0 : row0n, more ? r0 + 2 * C : 0);
            asm volatile("" ::: "memory");
            s = sum(qb, r0 + C, K, s);
        }
        const int kk = 64 * j + lane;
        if (kk < nchild) {
            const double tk = FLAT ? 0.0 : a.child_w[it.kbeg + kk].t;
            const double lam = (FLAT ? lam0 : sl_baseline(a, c, tk)) + s;
            prod *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
            pexp += __builtin_amdgcn_frexp_exp(lam);
            if (lambda_out) lambda_out[a.child_w[it.kbeg + kk].idx] = lam;           // (total_intensity, the two-pass gradient's 1/λ)
        }
        pexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
        j = jn; row0 = row0n; K = Kn;
    }
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;
    if (prod == 0.0) acc = -__builtin_inf();
    double blk = acc, blk_int = integ;
    nhp_block_sum2_n<NW>(blk, blk_int, red);
    if (!out) {
        if (tid == 0) { partials[2 * (size_t)blockIdx.x] = blk; partials[2 * (size_t)blockIdx.x + 1] = blk_int; }
        return;
    }
    int *flag = reinterpret_cast<int *>(red + 32);                  // the fused second stage of k_windowed_slices
    if (tid == 0) {
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[2 * (size_t)blockIdx.x + 1], blk_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, sh = blockIdx.x % NHP_SL_SHARDS;
        const unsigned int pop = (nb - sh + NHP_SL_SHARDS - 1) / NHP_SL_SHARDS;
        const unsigned int used = nb < NHP_SL_SHARDS ? nb : NHP_SL_SHARDS;
        int last = 0;
        if (__hip_atomic_fetch_add(&counter[32 * (1 + sh)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1)
            last = __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1;
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    double sl_ = 0.0, si = 0.0;
    for (unsigned int i = tid; i < gridDim.x; i += BLOCK) {
        sl_ += __hip_atomic_load(&partials[2 * (size_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si += __hip_atomic_load(&partials[2 * (size_t)i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nhp_block_sum2_n<NW>(sl_, si, red);
    if (tid == 0) *out = (0.0 - si) + sl_;
    for (int i = tid; i <= NHP_SL_SHARDS; i += BLOCK)
        __hip_atomic_store(&counter[32 * i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static nhp_slices slices_view(const nhp_cont_dataset *ds)
{
    nhp_slices sl;
    sl.row = ds->d_sl_row; sl.item0 = ds->d_sl_item0; sl.lo = ds->d_sl_lo; sl.hi = ds->d_sl_hi;
    sl.nsh = 16 - ds->sl_nb;
    sl.dmask = (1u << sl.nsh) - 1u;
    sl.dbits = 32 + sl.nsh;
    return sl;
}

// the planes, made at the first evaluation that wants them (data only)
static nhp_status ensure_slices(nhp_ctx *ctx, const nhp_cont_dataset *cds, const nhp_cont_args &a)
{
    if (cds->d_sl_lo) return NHP_OK;
    nhp_cont_dataset *ds = const_cast<nhp_cont_dataset *>(cds);
    const size_t n = ((size_t)ds->sl_rows + 16) * 64;
    if (hipMalloc((void **)&ds->d_sl_lo, 4 * n) != hipSuccess || hipMalloc((void **)&ds->d_sl_hi, 2 * n) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ds->d_sl_lo); (void)hipFree(ds->d_sl_hi);
        ds->d_sl_lo = nullptr; ds->d_sl_hi = nullptr;
        nhp_set_error(ctx, "out of device memory (child slices)");
        return NHP_ENOMEM;
    }
    NHP_HIP(ctx, hipMemsetAsync(ds->d_sl_lo + (size_t)ds->sl_rows * 64, 0, 4 * 16 * 64, ctx->stream));
    NHP_HIP(ctx, hipMemsetAsync(ds->d_sl_hi + (size_t)ds->sl_rows * 64, 0, 2 * 16 * 64, ctx->stream));
    const nhp_slices sl = slices_view(ds);
    if (ds->n_items > 0)
        hipLaunchKernelGGL(k_slices_build, dim3((unsigned)ds->n_items), dim3(256), 0, ctx->stream, a, sl, ds->d_sl_lo, ds->d_sl_hi);
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

static nhp_pslices pslices_view(const nhp_cont_dataset *ds)
{
    nhp_pslices ps;
    ps.row = ds->d_ps_row; ps.perm = ds->d_ps_perm; ps.lo = ds->d_ps_lo; ps.hi = ds->d_ps_hi;
    ps.spi = ds->ps_spi;
    ps.psh = 16 - ds->ps_sb;
    ps.dmask = (1u << ps.psh) - 1u;
    ps.dbits = 32 + ps.psh;
    ps.max_item = ds->max_item;
    ps.direct = 0;
    ps.grad = nullptr;
    return ps;
}

// The parent slices, made at the first gradient that wants them (data only): pair counts by (item, parent node) -> the
// lanes' order and every slice's rows (device) -> row offsets (host prefix sum over n_items·spi numbers: the one
// synchronisation) -> fill through LDS tickets -> every lane sorted.
static nhp_status ensure_parent_slices(nhp_ctx *ctx, const nhp_cont_dataset *cds, const nhp_cont_args &a)
{
    if (cds->d_ps_lo) return NHP_OK;
    nhp_cont_dataset *ds = const_cast<nhp_cont_dataset *>(cds);
    const int N = ds->N, spi = (N + 63) / 64, ni = ds->n_items;
    int sb = 0;
    while (((int64_t)1 << sb) <= ds->max_item) ++sb;               // bit length of max_item: the padding records' slot
    if (sb > 16 || 4 * (size_t)N > 160 * 1024) { nhp_set_error(ctx, "parent slices: item or node count out of range"); return NHP_ENOTIMPL; }
    uint32_t *d_cnt = nullptr, *d_rows = nullptr;
    uint16_t *d_rankof = nullptr;
    auto fail = [&](nhp_status st, const char *what) {
        (void)hipGetLastError();
        (void)hipFree(d_cnt); (void)hipFree(d_rows); (void)hipFree(d_rankof);
        (void)hipFree(ds->d_ps_row); (void)hipFree(ds->d_ps_perm); (void)hipFree(ds->d_ps_lo); (void)hipFree(ds->d_ps_hi);
        ds->d_ps_row = nullptr; ds->d_ps_perm = nullptr; ds->d_ps_lo = nullptr; ds->d_ps_hi = nullptr;
        nhp_set_error(ctx, "parent slices: %s", what);
        return st;
    };
    if (hipMalloc((void **)&d_cnt, 4 * (size_t)ni * N) != hipSuccess || hipMalloc((void **)&d_rows, 4 * (size_t)ni * spi) != hipSuccess ||
        hipMalloc((void **)&d_rankof, 2 * (size_t)ni * N) != hipSuccess ||
        hipMalloc((void **)&ds->d_ps_perm, 2 * (size_t)ni * spi * 64) != hipSuccess ||
        hipMalloc((void **)&ds->d_ps_row, 4 * ((size_t)ni * spi + 1)) != hipSuccess)
        return fail(NHP_ENOMEM, "out of device memory");
    const size_t lds = 4 * (size_t)N;
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_ps_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_ps_rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_ps_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k_ps_count, dim3((unsigned)ni), dim3(256), lds, ctx->stream, a, d_cnt);
    hipLaunchKernelGGL(k_ps_rank, dim3((unsigned)ni), dim3(256), lds, ctx->stream, N, spi, d_cnt, ds->d_ps_perm, d_rankof, d_rows);
    if (hipGetLastError() != hipSuccess) return fail(NHP_EHIP, "launch failed");
    std::vector<uint32_t> rows((size_t)ni * spi), off((size_t)ni * spi + 1);
    if (hipMemcpyAsync(rows.data(), d_rows, 4 * rows.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(NHP_EHIP, "copy failed");
    uint64_t run = 0;
    for (size_t q = 0; q < rows.size(); ++q) { off[q] = (uint32_t)run; run += rows[q]; }
    off[rows.size()] = (uint32_t)run;
    if (run >= ((uint64_t)1 << 25)) return fail(NHP_ENOTIMPL, "too many rows");
    ds->ps_rows = (int64_t)run; ds->ps_spi = spi; ds->ps_sb = sb;
    const size_t n = ((size_t)run + 16) * 64;
    if (hipMalloc((void **)&ds->d_ps_lo, 4 * n) != hipSuccess || hipMalloc((void **)&ds->d_ps_hi, 2 * n) != hipSuccess)
        return fail(NHP_ENOMEM, "out of device memory");
    if (hipMemcpyAsync(ds->d_ps_row, off.data(), 4 * off.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(ds->d_ps_lo + (size_t)run * 64, 0, 4 * 16 * 64, ctx->stream) != hipSuccess ||
        hipMemsetAsync(ds->d_ps_hi + (size_t)run * 64, 0, 2 * 16 * 64, ctx->stream) != hipSuccess)
        return fail(NHP_EHIP, "copy failed");
    const nhp_pslices ps = pslices_view(ds);
    hipLaunchKernelGGL(k_ps_fill, dim3((unsigned)ni), dim3(256), lds, ctx->stream, a, ps, d_cnt, d_rankof, ds->d_ps_lo, ds->d_ps_hi);
    hipLaunchKernelGGL(k_ps_sort, dim3((unsigned)ni), dim3(256), 0, ctx->stream, N, ps, d_cnt, ds->d_ps_lo, ds->d_ps_hi);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(NHP_EHIP, "fill failed");   // (`off` and the scratch go out of scope)
    (void)hipFree(d_cnt); (void)hipFree(d_rows); (void)hipFree(d_rankof);
    return NHP_OK;
}

// grad != nullptr: log-likelihood AND gradient (k_windowed_slices<.., GRAD>); the gradient must have been initialised by
// k_grad_init unless `direct`
static nhp_status launch_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int mask_integral, double *d_out,
                                double *d_grad, bool direct, bool *launched)
{
    *launched = false;
    if (!ds->d_sl_row || ds->n_items <= 0 || m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_OK;
    if (getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) return NHP_OK;          // (A/B switch, read per call: the tests flip it)
    const size_t lds = 320 + 16 * ((size_t)ds->N + 1) + 512 + (d_grad ? 8 * ((size_t)ds->max_item + 1) : 0);
    if (lds > 160 * 1024) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items));
    nhp_cont_args a = nhp_make_args(ds, m);
    NHP_TRY(ensure_slices(ctx, ds, a));
    const nhp_slices sl = slices_view(ds);
    nhp_pslices ps{};
    if (d_grad) {
        NHP_TRY(ensure_parent_slices(ctx, ds, a));
        ps = pslices_view(ds);
        ps.grad = d_grad;
        ps.direct = direct ? 1 : 0;
    }
    // waves per workgroup from the slices an item has (one item per node at N >= 1024: ~15 slices; short items at small N);
    // rows per request from the windows' length.  NHP_SLICES_CFG = "BLOCK,C" overrides (tools/dbg/slicesweep.sh)
    const int per_item = (ds->max_item + 63) / 64;
    // (rows in flight per wave = 2C: short slices are served by L2 / the Infinity Cache and want few, long ones stream from HBM)
    const double mean_rows = ds->n_slices > 0 ? (double)ds->sl_rows / (double)ds->n_slices : 0.0;
    int B = per_item >= 12 ? 512 : per_item >= 6 ? 256 : per_item >= 3 ? 128 : 64, C = mean_rows >= 48.0 ? 8 : mean_rows >= 16.0 ? 4 : 2;
    if (const char *cfg = getenv("NHP_SLICES_CFG")) sscanf(cfg, "%d,%d", &B, &C);
    dim3 grid((unsigned)ds->n_items);
    bool ok = false;
    const bool flat = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS;
#define NHP_SLAUNCH(b, cc, f, g)                                                                                      \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void *)k_windowed_slices<b, cc, f, g>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed_slices<b, cc, f, g>), grid, dim3(b), lds, ctx->stream, a, sl, ps, mask_integral, ctx->d_partials, \
                           ctx->d_counter, d_out);                                                                    \
    } while (0)
#define NHP_SCASE(b, cc)                                                                                              \
    if (!ok && B == b && C == cc) {                                                                                   \
        ok = true;                                                                                                    \
        if (d_grad) { if (flat) NHP_SLAUNCH(b, cc, true, true); else NHP_SLAUNCH(b, cc, false, true); }               \
        else { if (flat) NHP_SLAUNCH(b, cc, true, false); else NHP_SLAUNCH(b, cc, false, false); }                    \
    }
#define NHP_SROW(b) NHP_SCASE(b, 2) NHP_SCASE(b, 4) NHP_SCASE(b, 8)
    NHP_SROW(64) NHP_SROW(128) NHP_SROW(256) NHP_SROW(512) NHP_SROW(1024)
    if (!ok) { B = 256; C = 4; NHP_SCASE(256, 4) }
#undef NHP_SLAUNCH
#undef NHP_SROW
#undef NHP_SCASE
    NHP_HIP(ctx, hipGetLastError());
    *launched = true;
    return NHP_OK;
}

nhp_status nhp_launch_windowed_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int mask_integral,
                                      double *d_out, bool *launched)
{
    return launch_slices(ctx, ds, m, mask_integral, d_out, nullptr, false, launched);
}

static nhp_status ensure_lq_planes(nhp_ctx *ctx, nhp_cont_dataset *ds, const nhp_cont_args &a, const nhp_slices &sl, bool *ok);

// logit-normal impulses: the same call for k_windowed_slices_ln (NHP_SLICES_LN=0: the pair-cache kernel)
nhp_status nhp_launch_windowed_slices_ln(nhp_ctx *ctx, const nhp_cont_dataset *cds, const nhp_cont_model *m, int mask_integral,
                                         double *d_out, double *d_lambda, bool *launched)
{
    *launched = false;
    if (!cds->d_sl_row || cds->n_items <= 0 || m->impulse_kind != NHP_IMPULSE_LOGITNORMAL) return NHP_OK;
    if ((getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) || (getenv("NHP_SLICES_LN") && atoi(getenv("NHP_SLICES_LN")) == 0)) return NHP_OK;
    const size_t lds = 320 + 24 * ((size_t)cds->N + 1);
    if (lds > 160 * 1024) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)cds->n_items));
    nhp_cont_args a = nhp_make_args(cds, m);
    NHP_TRY(ensure_slices(ctx, cds, a));
    nhp_cont_dataset *ds = const_cast<nhp_cont_dataset *>(cds);
    const nhp_slices sl = slices_view(ds);
    bool planes = false;
    NHP_TRY(ensure_lq_planes(ctx, ds, a, sl, &planes));
    if (!planes) return NHP_OK;
    const int per_item = (ds->max_item + 63) / 64;
    const double mean_rows = ds->n_slices > 0 ? (double)ds->sl_rows / (double)ds->n_slices : 0.0;
    // (metric size, µs per evaluation: 512 threads x 2 rows 35.2, 512 x 4 37.0, 256 x 2 34.3, 256 x 4 36.8, 64 x 2 48.9; the pair cache 39.5)
    int B = per_item >= 3 ? 256 : 64, C = mean_rows >= 32.0 ? 4 : 2;
    if (const char *cfg = getenv("NHP_SLICES_LN_CFG")) sscanf(cfg, "%d,%d", &B, &C);
    if (!((B == 64 || B == 256 || B == 512) && (C == 2 || C == 4))) { B = 256; C = 2; }
    const bool flat = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS;
    dim3 grid((unsigned)ds->n_items);
#define NHP_LNL(b, cc, f)                                                                                             \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void *)k_windowed_slices_ln<b, cc, f>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed_slices_ln<b, cc, f>), grid, dim3(b), lds, ctx->stream, a, sl, (const double *)ds->d_sl_L,  \
                           (const double *)ds->d_sl_Q, mask_integral, ctx->d_partials, ctx->d_counter, d_out, d_lambda); \
    } while (0)
#define NHP_LNC(b, cc) do { if (flat) NHP_LNL(b, cc, true); else NHP_LNL(b, cc, false); } while (0)
    if (B == 64) { if (C == 2) NHP_LNC(64, 2); else NHP_LNC(64, 4); }
    else if (B == 256) { if (C == 2) NHP_LNC(256, 2); else NHP_LNC(256, 4); }
    else { if (C == 2) NHP_LNC(512, 2); else NHP_LNC(512, 4); }
#undef NHP_LNC
#undef NHP_LNL
    NHP_HIP(ctx, hipGetLastError());
    *launched = true;
    return NHP_OK;
}

// Log-likelihood -> *d_out and gradient -> d_grad [P] of the dataset's own windows (mask_integral = 1: the windowed route's
// masked integral).  When every item is its node's only one, the dataset is whole and the baseline flat, the kernel stores
// every entry of the gradient itself; otherwise k_grad_init (cont_grad.hip) must have run on d_grad (*needs_init).
nhp_status nhp_launch_grad_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out, double *d_grad,
                                  bool *launched)
{
    *launched = false;
    if (getenv("NHP_GRAD_SLICES") && atoi(getenv("NHP_GRAD_SLICES")) == 0) return NHP_OK;   // (A/B switch: the two-pass route of cont_grad.hip)
    const bool direct = ds->all_sole && !nhp_is_column_shard(ds) && m->baseline_kind == NHP_BASELINE_HOMOGENEOUS;
    return launch_slices(ctx, ds, m, 1, d_out, d_grad, direct, launched);
}

bool nhp_grad_slices_direct(const nhp_cont_dataset *ds, const nhp_cont_model *m)
{
    return ds->all_sole && !nhp_is_column_shard(ds) && m->baseline_kind == NHP_BASELINE_HOMOGENEOUS;
}

// diagnostics (tools/dbg): the parent-slice planes as they sit on the device; returns the number of records (rows * 64), or -1
extern "C" int64_t nhp_debug_parent_slices(const nhp_cont_dataset *ds, uint32_t *lo, uint16_t *hi, int64_t cap)
{
    if (!ds || !ds->d_ps_lo) return -1;
    const int64_t n = ds->ps_rows * 64;
    if (lo && hi && cap >= n) {
        if (hipMemcpy(lo, ds->d_ps_lo, 4 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -2;
        if (hipMemcpy(hi, ds->d_ps_hi, 2 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    }
    return n;
}

template <int S>
static nhp_status launch_sets(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *const *ms, int32_t slot0, size_t lds)
{
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items * S));
    nhp_cont_args a = nhp_make_args(ds, ms[0]);
    NHP_TRY(ensure_slices(ctx, ds, a));
    const nhp_slices sl = slices_view(ds);
    nhp_sets st;
    for (int k = 0; k < NHP_SETS_MAX; ++k) {
        const nhp_cont_model *m = ms[k < S ? k : 0];
        st.p1[k] = m->d_p1; st.W[k] = m->d_W; st.A[k] = m->has_A ? m->d_A : nullptr; st.lambda0[k] = m->d_lambda0;
        st.out[k] = ctx->d_results + slot0 + (k < S ? k : 0);
    }
    const double mean_rows = ds->n_slices > 0 ? (double)ds->sl_rows / (double)ds->n_slices : 0.0;
    dim3 grid((unsigned)ds->n_items);
    // (the columns of S models leave room for two workgroups of 512 per CU at N = 1024, S = 4; NHP_SETS_CFG = "BLOCK,C" overrides)
    int B = (ds->max_item + 63) / 64 >= 6 ? 512 : 256, C = mean_rows >= 16.0 ? 4 : 2;
    if (const char *cfg = getenv("NHP_SETS_CFG")) sscanf(cfg, "%d,%d", &B, &C);
#define NHP_BL(b, cc)                                                                                                  \
    do {                                                                                                               \
        if (lds > 64 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void *)k_slices_batch<b, cc, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_slices_batch<b, cc, S>), grid, dim3(b), lds, ctx->stream, a, sl, st, ctx->d_partials, ctx->d_counter); \
    } while (0)
    if (B == 256 && C == 2) NHP_BL(256, 2); else if (B == 256) NHP_BL(256, 4);
    else if (B == 1024 && C == 2) NHP_BL(1024, 2); else if (B == 1024) NHP_BL(1024, 4);
    else if (C == 2) NHP_BL(512, 2); else NHP_BL(512, 4);
#undef NHP_BL
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

nhp_status nhp_launch_slices_batch(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *const *ms, int S, int32_t slot0,
                                   bool *launched)
{
    *launched = false;
    if (!ds->d_sl_row || ds->n_items <= 0 || (S != 2 && S != 4)) return NHP_OK;
    if ((getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) || (getenv("NHP_BATCH_SLICES") && atoi(getenv("NHP_BATCH_SLICES")) == 0)) return NHP_OK;
    for (int k = 0; k < S; ++k)
        if (!ms[k] || ms[k]->impulse_kind != NHP_IMPULSE_EXPONENTIAL || ms[k]->baseline_kind != NHP_BASELINE_HOMOGENEOUS) return NHP_OK;
    const size_t lds = 320 + 16 * ((size_t)ds->N + 1) * (size_t)S + 512;
    if (lds > 160 * 1024) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    if (S == 4) NHP_TRY(launch_sets<4>(ctx, ds, ms, slot0, lds)); else NHP_TRY(launch_sets<2>(ctx, ds, ms, slot0, lds));
    *launched = true;
    return NHP_OK;
}

// the planes of logit(x) and 1/(x(1-x)), made at the first call that wants them (data only); *ok = false: no room
static nhp_status ensure_lq_planes(nhp_ctx *ctx, nhp_cont_dataset *ds, const nhp_cont_args &a, const nhp_slices &sl, bool *ok)
{
    *ok = true;
    if (ds->d_sl_L) return NHP_OK;
    const size_t n = ((size_t)ds->sl_rows + 16) * 64;
    if (hipMalloc((void **)&ds->d_sl_L, 8 * n) != hipSuccess || hipMalloc((void **)&ds->d_sl_Q, 8 * n) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(ds->d_sl_L); (void)hipFree(ds->d_sl_Q);
        ds->d_sl_L = nullptr; ds->d_sl_Q = nullptr;
        *ok = false;
        return NHP_OK;
    }
    NHP_HIP(ctx, hipMemsetAsync(ds->d_sl_L + (size_t)ds->sl_rows * 64, 0, 8 * 16 * 64, ctx->stream));
    NHP_HIP(ctx, hipMemsetAsync(ds->d_sl_Q + (size_t)ds->sl_rows * 64, 0, 8 * 16 * 64, ctx->stream));
    hipLaunchKernelGGL(k_slices_build_lq, dim3((unsigned)ds->n_items), dim3(256), 0, ctx->stream, a, sl, ds->d_sl_L, ds->d_sl_Q);
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

nhp_status nhp_launch_sampler_slices(nhp_ctx *ctx, const nhp_cont_dataset *cds, const nhp_cont_model *m, const double *d_u, uint64_t seed,
                                     uint64_t step, int64_t *parents, int64_t *pnodes, int32_t *pn_b, double *dt_b, int *d_err, bool *launched)
{
    *launched = false;
    // (only windows below Julia's pairwise-sum threshold: the slice kernel sums sequentially)
    const bool expo = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    if (!cds->d_sl_row || cds->n_items <= 0 || (!expo && m->impulse_kind != NHP_IMPULSE_LOGITNORMAL) || cds->sl_max_rows + 1 > 1024) return NHP_OK;
    if ((getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) || (getenv("NHP_SAMPLER_SLICES") && atoi(getenv("NHP_SAMPLER_SLICES")) == 0)) return NHP_OK;
    int B = 512, CACHE = 8;
    if (const char *cfg = getenv("NHP_SAMPLER_CFG")) sscanf(cfg, "%d,%d", &B, &CACHE);
    if (!((B == 256 || B == 512) && (CACHE == 8 || CACHE == 16))) { B = 512; CACHE = 8; }
    const size_t lds = 24 * ((size_t)cds->N + 1) + 8 * (size_t)(B / 64) * CACHE * 64;
    if (lds > 160 * 1024) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    nhp_cont_args a = nhp_make_args(cds, m);
    NHP_TRY(ensure_slices(ctx, cds, a));
    nhp_cont_dataset *ds = const_cast<nhp_cont_dataset *>(cds);
    const nhp_slices sl = slices_view(ds);
    if (expo && !ds->d_sl_D) {
        const size_t n = ((size_t)ds->sl_rows + 16) * 64;
        if (hipMalloc((void **)&ds->d_sl_D, 8 * n) != hipSuccess) { (void)hipGetLastError(); ds->d_sl_D = nullptr; return NHP_OK; }
        NHP_HIP(ctx, hipMemsetAsync(ds->d_sl_D + (size_t)ds->sl_rows * 64, 0, 8 * 16 * 64, ctx->stream));
        hipLaunchKernelGGL(k_slices_build_d, dim3((unsigned)ds->n_items), dim3(256), 0, ctx->stream, a, sl, ds->d_sl_D);
        NHP_HIP(ctx, hipGetLastError());
    }
    if (!expo) {
        bool planes = false;
        NHP_TRY(ensure_lq_planes(ctx, ds, a, sl, &planes));
        if (!planes) return NHP_OK;                                 // no room: the caller keeps its other kernel
    }
    dim3 grid((unsigned)ds->n_items);
#define NHP_SAMP(b, cc)                                                                                                \
    do {                                                                                                               \
        if (expo) {                                                                                                    \
            if (lds > 64 * 1024)                                                                                       \
                (void)hipFuncSetAttribute((const void *)k_sampler_slices<b, cc, NHP_IMPULSE_EXPONENTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_sampler_slices<b, cc, NHP_IMPULSE_EXPONENTIAL>), grid, dim3(b), lds, ctx->stream, a, sl, (const double *)ds->d_sl_D, \
                               (const double *)nullptr, d_u, seed, step, parents, pnodes, pn_b, dt_b, d_err);      \
        } else {                                                                                                       \
            if (lds > 64 * 1024)                                                                                       \
                (void)hipFuncSetAttribute((const void *)k_sampler_slices<b, cc, NHP_IMPULSE_LOGITNORMAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_sampler_slices<b, cc, NHP_IMPULSE_LOGITNORMAL>), grid, dim3(b), lds, ctx->stream, a, sl, (const double *)ds->d_sl_L, \
                               (const double *)ds->d_sl_Q, d_u, seed, step, parents, pnodes, pn_b, dt_b, d_err);      \
        }                                                                                                              \
    } while (0)
    if (B == 256 && CACHE == 8) NHP_SAMP(256, 8); else if (B == 256) NHP_SAMP(256, 16);
    else if (CACHE == 8) NHP_SAMP(512, 8); else NHP_SAMP(512, 16);
#undef NHP_SAMP
    NHP_HIP(ctx, hipGetLastError());
    *launched = true;
    return NHP_OK;
}
