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

#include "nhp_internal.h"
#include "nhp_math.h"

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

#ifdef NHP_STAMP      // diagnostic build only (tools/dbg/slstamps.py): s_memtime at the phase boundaries of wave 0 of every workgroup
__device__ unsigned long long g_sl_stamps[8 * 4096];
#define NHP_SL_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_sl_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
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

// C rows of a slice are requested at a time into one of two register sets: the next set is in flight while this one is
// summed, across slice boundaries too (the first rows of a wave's next slice are requested under the last rows of this one,
// and the very first set before the column is staged: its addresses need the slice table only).
template <int BLOCK, int C, bool FLAT>
__global__ __launch_bounds__(BLOCK) void k_windowed_slices(nhp_cont_args a, nhp_slices sl, int mask_integral,
                                                            double *__restrict__ partials, unsigned int *__restrict__ counter,
                                                            double *__restrict__ out)
{
    constexpr int NW = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [2 * NW <= 32] + flag at [32]
    double2 *col = reinterpret_cast<double2 *>(smem + 320);         // [N + 1] {-θ·unit·64/ln2, a·w·θ}; [N] = {0, 0}
    double *etab = reinterpret_cast<double *>(col + a.N + 1);       // [64] 2^(j/64)
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
    for (int p0 = 0; p0 < N || p0 == 0; p0 += UN * BLOCK) {
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
            request(qa, row0, 0);
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
    if (out && it.first) integ += sl_baseline_integral_col(a, c);
    __syncthreads();
    NHP_SL_STAMP(1);

    const double lam0 = FLAT ? a.lambda0[c] : 0.0;
    const uint32_t dmask = sl.dmask;
    const int nsh = sl.nsh;
    auto term = [&](const uint32_t lo, const uint32_t h) {
        // the high delay bits under the exponent of 2^52 (one v_bfi_b32): the double 2^52 + delay
        uint32_t hw;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hw) : "s"(dmask), "v"(h), "v"(0x43300000u));
        const double v = __hiloint2double((int)hw, (int)lo) - 4503599627370496.0;
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
            const double lam = (FLAT ? lam0 : sl_baseline(a, c, a.child_w[it.kbeg + kk].t)) + s;
            prod *= lam < 0.0 ? __builtin_nan("") : __builtin_amdgcn_frexp_mant(lam);
            pexp += __builtin_amdgcn_frexp_exp(lam);
        }
        pexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
        j = jn; row0 = row0n; K = Kn;
    }
    NHP_SL_STAMP(2);
    double acc = nhp_log(prod) + (double)pexp * 6.93147180559945286e-01;
    if (prod == 0.0) acc = -__builtin_inf();
    double blk = acc, blk_int = integ;
    nhp_block_sum2_n<NW>(blk, blk_int, red);
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

nhp_status nhp_launch_windowed_slices(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int mask_integral,
                                      double *d_out, bool *launched)
{
    *launched = false;
    if (!ds->d_sl_row || ds->n_items <= 0 || m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_OK;
    if (getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) return NHP_OK;          // (A/B switch, read per call: the tests flip it)
    const size_t lds = 320 + 16 * ((size_t)ds->N + 1) + 512;
    if (lds > 160 * 1024) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->n_items));
    nhp_cont_args a = nhp_make_args(ds, m);
    NHP_TRY(ensure_slices(ctx, ds, a));
    const nhp_slices sl = slices_view(ds);
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
#define NHP_SLAUNCH(b, cc, f)                                                                                         \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            (void)hipFuncSetAttribute((const void *)k_windowed_slices<b, cc, f>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_windowed_slices<b, cc, f>), grid, dim3(b), lds, ctx->stream, a, sl, mask_integral, ctx->d_partials, \
                           ctx->d_counter, d_out);                                                                    \
    } while (0)
#define NHP_SCASE(b, cc)                                                                                              \
    if (!ok && B == b && C == cc) {                                                                                   \
        ok = true;                                                                                                    \
        if (flat) NHP_SLAUNCH(b, cc, true); else NHP_SLAUNCH(b, cc, false);                                           \
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
