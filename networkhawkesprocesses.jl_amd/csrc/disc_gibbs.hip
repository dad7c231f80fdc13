// Discrete-time Gibbs sampling on the GPU: parent counts of a sweep and the adjacency-matrix
// sweep of the network process.  The shared pieces -- dataset, bump table staging, the intensity GEMM -- live
// in disc.hip (declared in nhp_internal.h).
#include <math.h>

#include <algorithm>
#include <vector>

#include "nhp_internal.h"
#include "nhp_math.h"
#include "nhp_rng.h"

// ---- discrete Gibbs parent counts (reference resample_parents / resample_parent
// src/parents.jl:82-116 reduced over time to counts[c + N·k] = Σ_t parents[t, c, k], which is all the
// discrete resample! methods read: src/baselines.jl:413-419, src/weights.jl:28-35,
// src/impulses.jl:337-353).  Only occupied bins draw anything, and a bin's Multinomial(n, μ) is n
// categorical draws over 1 + N·B categories, i.e. an inverse-CDF walk along the bin's row of
// Z = base ⊕ G·E.  The reference materialises parents[T, N, 1+NB]; here a workgroup takes a 64-bin x
// 128-node tile, lists its occupied bins, and walks the category axis twice in chunks staged through
// LDS (G rows and E rows shared by all the tile's bins): once for the row total, once comparing the
// running sum with the bin's ascending thresholds u_(1) < u_(2) < ... (order statistics generated one
// at a time from Philox, so n events cost one walk).  One lane owns a bin's running sum, in the
// reference's category order with separate multiply and add: counts equal the oracle's, bit for bit.
#ifndef RP_KC
#define RP_KC 16        // categories a chunk (variant builds: 8 / 32, tools/dbg notes)
#endif
#ifndef RP_ABL
#define RP_ABL 0        // (timing ablations, wrong results: 1 = no arithmetic in the walks, 2 = no global loads of the chunks, 3 = conflict-free G reads)
#endif
#define RP_KEY 0xD15C0DE5EEDC0FFEull
#ifdef RP_STAMP
// (debug build, tools/dbg/rpstamps.py) per workgroup: s_memtime ticks spent, over the chunks of walk 1 [0..3] and walk 2 [4..7], in the
// barrier before a chunk is staged, the wait for its loads, staging + barrier, and its arithmetic
__device__ unsigned long long g_rp_stamps[8 * 4096];
extern "C" int nhp_debug_rp_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rp_stamps), sizeof(unsigned long long) * (size_t)n);
}
#define RP_T() __builtin_amdgcn_s_memtime()
#define RP_ST_A() const unsigned long long ta = RP_T(); __syncthreads(); const unsigned long long tb0 = RP_T(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long tb = RP_T()
#define RP_ST_C() const unsigned long long tc = RP_T()
#define RP_ST_D(w) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); { const unsigned long long td = RP_T(); st[4 * (w)] += tb0 - ta; st[4 * (w) + 1] += tb - tb0; st[4 * (w) + 2] += tc - tb; st[4 * (w) + 3] += td - tc; }
#else
#define RP_ST_A() __syncthreads()
#define RP_ST_C() do { } while (0)
#define RP_ST_D(w) do { } while (0)
#endif

__device__ __attribute__((noinline)) double rp_next_u(double u_prev, int remaining, uint64_t seed, uint64_t step, uint64_t bin, int j)
{
#pragma clang fp contract(off)
    const double V = nhp_philox_uniform(seed ^ RP_KEY, step, (bin << 20) | (uint64_t)j);
    const double r = nhp_exp(nhp_log(V) / (double)remaining);
    const double w = 1.0 - r;
    return u_prev + (1.0 - u_prev) * w;
}

// RP_SLOTS = occupied bins a thread carries through one pair of walks: the host picks the smallest of 1, 2, 4 whose
// RP_TH·RP_SLOTS slots hold a tile's occupied bins (an empty slot costs the walk as much as a full one: at config 4 a
// 64 x 128 tile holds ~400, and two slots per thread take 14.7 ms where four took 20.9).
// RP_TT x RP_CT = the tile (bins x child nodes), RP_TH threads.  What a launch moves is the staging traffic: per chunk of RP_KC
// categories a tile fetches (RP_TT + RP_CT)·RP_KC doubles for RP_TT·RP_CT·rate occupied bins -- 78 GB per sweep at config 4 with
// 64 x 128 tiles (5.6 TB/s out of L2 / the Infinity Cache: the bound of the 14 ms launch), half of that with 128 x 256.
// COLM: the list runs column by column, every column padded to a multiple of RP_SLOTS, and a thread's slots are consecutive
// entries -- bins of ONE node: E[category, node] is read once per category for all of them (and consecutive lanes read
// consecutive nodes: no bank conflicts there), which leaves (1 + SLOTS) / SLOTS LDS reads per multiply-add instead of 2.
// CHK: walk 1 keeps the running sum at the start of each eighth of the category axis (registers); walk 2 then enters a bin's
// chain at the eighth its first threshold falls into -- the same partial sum, bit for bit -- and leaves it when the bin's
// events are placed: a bin with one event (97 % of them at 5 % occupancy) reads a sixteenth of the categories instead of all.
// XCHG (with CHK): between the walks the bins still to be placed change lanes through LDS, ordered by the eighth their chain is
// entered at -- a wave then holds bins of one or two eighths and skips the chunks of the others whole (scattered, some lane of
// nearly every wave was inside its range in every chunk: the skipping of CHK was per lane only).
template <int RP_SLOTS, int RP_TT, int RP_CT, int RP_TH, bool COLM, bool CHK, bool XCHG>
__global__ __launch_bounds__(RP_TH, 4) void k_disc_resample_parents(const double *__restrict__ dataT, const double *__restrict__ conv,
                                                               const double *__restrict__ E2, const double *__restrict__ base,
                                                               const double *__restrict__ baseT, int64_t T, int N, int B,
                                                               unsigned b_magic, uint64_t seed, uint64_t step,
                                                               int *__restrict__ counts, int *__restrict__ base_counts, int xcd_ncy)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char rp_smem[];
    double (*Gt)[RP_TT] = reinterpret_cast<double (*)[RP_TT]>(rp_smem);                                   // [RP_KC][RP_TT]
    double (*Et)[RP_CT + 1] = reinterpret_cast<double (*)[RP_CT + 1]>(rp_smem + 8 * RP_KC * RP_TT);      // [RP_KC][RP_CT + 1]
    unsigned short *list = reinterpret_cast<unsigned short *>(rp_smem + 8 * RP_KC * (RP_TT + RP_CT + 1)); // [RP_TT * RP_CT]
    __shared__ int nb, wcnt[RP_TH / 64];
    const int tid = threadIdx.x, K = N * B;
    // Tile of a workgroup.  The node tiles of one bin range read the same rows of the 3.3 GB convolution; dispatched a grid
    // row apart they each fetch them from HBM (the matrix does not stay in the Infinity Cache: 4 x 3.3 GB per walk with four
    // node tiles).  xcd > 0: eight consecutive bin ranges x all node tiles share 8·ncy consecutive workgroup ids, node tile
    // cy of bin range i at id cy·8 + i -- the siblings are dispatched together and, ids being dealt round-robin to the 8 XCDs,
    // land on the same XCD: one of them misses, the others hit its L2.
    int tx = blockIdx.x, cy = blockIdx.y;
    if (xcd_ncy > 0) {
        const int g8 = blockIdx.x / (8 * xcd_ncy), r = blockIdx.x % (8 * xcd_ncy);
        cy = r / 8; tx = g8 * 8 + r % 8;
        if ((int64_t)tx * RP_TT >= T) return;
    }
    const int64_t t0 = (int64_t)tx * RP_TT;
    const int c0 = cy * RP_CT;
    // Occupied bins, listed bin-row by bin-row (entry = tl·RP_CT + cl): consecutive lanes then share a
    // bin row, so a wave's reads of a G row collapse to a few broadcast addresses and its reads of an E
    // row hit distinct banks.  Flags are gathered with coalesced loads (t fastest), then compacted in order.
    unsigned char *occ = reinterpret_cast<unsigned char *>(&Et[0][0]);       // [RP_TT][RP_CT] (COLM: [RP_CT][RP_TT]), before Et is used
    for (int i = tid; i < RP_TT * RP_CT; i += RP_TH) {
        const int tl_ = i % RP_TT, cl_ = i / RP_TT;
        const int64_t t = t0 + tl_;
        const int c = c0 + cl_;
        occ[COLM ? i : tl_ * RP_CT + cl_] = (t < T && c < N && dataT[(size_t)t + (size_t)T * c] > 0.0) ? 1 : 0;
    }
    __syncthreads();
    if (COLM) {
        // whole columns per wave; a column's entries in bin order, then padding (0xFFFF) up to a multiple of RP_SLOTS
        static_assert(!COLM || (RP_TT % 64 == 0 && (RP_TT * RP_CT / (RP_TH / 64)) % RP_TT == 0 && RP_TT * RP_CT < 65535), "tile shape (column-major list)");
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int CPW = RP_CT / (RP_TH / 64);                  // columns per wave
        int cnt = 0;
        for (int cl_ = wave * CPW; cl_ < (wave + 1) * CPW; ++cl_) {
            int m = 0;
#pragma unroll
            for (int q = 0; q < RP_TT / 64; ++q) m += __popcll(__ballot(occ[cl_ * RP_TT + 64 * q + lane] != 0));
            cnt += (m + RP_SLOTS - 1) / RP_SLOTS * RP_SLOTS;
        }
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        for (int cl_ = wave * CPW; cl_ < (wave + 1) * CPW; ++cl_) {
            int m = 0;
#pragma unroll
            for (int q = 0; q < RP_TT / 64; ++q) {
                const bool f = occ[cl_ * RP_TT + 64 * q + lane] != 0;
                const unsigned long long bm = __ballot(f);
                if (f) list[off + m + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)((64 * q + lane) * RP_CT + cl_);
                m += __popcll(bm);
            }
            const int mp = (m + RP_SLOTS - 1) / RP_SLOTS * RP_SLOTS;
            if (lane < mp - m) list[off + m + lane] = 0xFFFFu;
            off += mp;
        }
        if (tid == RP_TH - 1) nb = off;
    } else {
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int PER_WAVE = RP_TT * RP_CT / (RP_TH / 64);
        int cnt = 0;
        for (int i = wave * PER_WAVE + lane; i < (wave + 1) * PER_WAVE; i += 64) cnt += __popcll(__ballot(occ[i] != 0));
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        for (int i = wave * PER_WAVE + lane; i < (wave + 1) * PER_WAVE; i += 64) {
            const bool f = occ[i] != 0;
            const unsigned long long m = __ballot(f);
            if (f) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            off += __popcll(m);
        }
        if (tid == RP_TH - 1) nb = off;
    }
    __syncthreads();
    const int nbins = nb;

    // A chunk of RP_KC categories is fetched into registers one chunk ahead (under the previous chunk's
    // arithmetic) and written to LDS between two barriers.
    constexpr int GN = RP_KC * RP_TT / RP_TH, EN = RP_KC * RP_CT / RP_TH;
    static_assert(GN >= 1 && EN >= 1 && RP_TT * RP_CT <= 65536 && RP_TT * RP_CT <= 8 * RP_KC * (RP_CT + 1), "tile shape");
    static_assert(RP_TH % RP_TT == 0 && RP_TH % RP_KC == 0, "staging coordinates below assume these");
    double rg[GN], re[EN];
    // staging coordinates are fixed per thread: G element r is (category row tid/RP_TT + (RP_TH/RP_TT)·r,
    // bin tid % RP_TT), E element r is (category tid % RP_KC, node tid/RP_KC + (RP_TH/RP_KC)·r).
    // Category q = p·B + b reads Ŝ[t, p, b]; q / B is a multiply-high by the host-made reciprocal.
    const int g_tt = tid % RP_TT, g_k0 = tid / RP_TT, e_kk = tid % RP_KC, e_c0 = tid / RP_KC;
    const bool g_ok = t0 + g_tt < T;
    const double *g_base = conv + (size_t)(g_ok ? t0 + g_tt : 0);
    const unsigned T32 = (unsigned)T;
    // loads are unconditional from clamped (always valid) addresses and zeroed by a select afterwards:
    // predicated loads compile to one exec-mask branch each, and twelve of them per chunk spill SGPRs
    unsigned gmask = 0, emask = 0;            // which of the fetched values are real (applied when staged,
    auto fetch = [&](int q0) {                // so that the loads stay in flight under the arithmetic)
        gmask = 0; emask = 0;
        if (RP_ABL == 2) return;
#pragma unroll
        for (int r = 0; r < GN; ++r) {
            const int qr = q0 + g_k0 + (RP_TH / RP_TT) * r;
            const unsigned q = (unsigned)(qr < K ? qr : K - 1);
            const unsigned pq = B == 1 ? q : __umulhi(q, b_magic), bq = q - pq * (unsigned)B;
            rg[r] = g_base[(size_t)T32 * (size_t)(pq + (unsigned)N * bq)];
            gmask |= (qr < K && g_ok ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < EN; ++r) {
            const int cr = c0 + e_c0 + (RP_TH / RP_KC) * r, qr = q0 + e_kk;
            re[r] = E2[(size_t)(qr < K ? qr : K - 1) + (size_t)(cr < N ? cr : N - 1) * K];
            emask |= (qr < K && cr < N ? 1u : 0u) << r;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int r = 0; r < GN; ++r) { const int e = tid + RP_TH * r; Gt[e / RP_TT][e % RP_TT] = (gmask >> r) & 1u ? rg[r] : 0.0; }
#pragma unroll
        for (int r = 0; r < EN; ++r) { const int e = tid + RP_TH * r; Et[e % RP_KC][e / RP_KC] = (emask >> r) & 1u ? re[r] : 0.0; }
    };

    for (int b0 = 0; b0 < nbins; b0 += RP_TH * RP_SLOTS) {
        int tl[RP_SLOTS], cl[RP_SLOTS], n[RP_SLOTS], j[RP_SLOTS];
        double cum[RP_SLOTS], total[RP_SLOTS], thr[RP_SLOTS], u[RP_SLOTS];
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s) {
            const int idx = COLM ? b0 + tid * RP_SLOTS + s : b0 + tid + RP_TH * s;
            const unsigned int e0 = idx < nbins ? list[idx] : 0xFFFFu;
            const bool real = !COLM ? idx < nbins : e0 != 0xFFFFu;
            const int e = real ? (int)e0 : 0;
            tl[s] = e / RP_CT; cl[s] = e % RP_CT;
            n[s] = real ? (int)dataT[(size_t)(t0 + tl[s]) + (size_t)T * (c0 + cl[s])] : 0;
            j[s] = 0;
            cum[s] = n[s] > 0 ? (baseT ? baseT[(size_t)(t0 + tl[s]) + (size_t)T * (c0 + cl[s])] : base[c0 + cl[s]]) : 0.0;
            total[s] = 0.0; thr[s] = 0.0; u[s] = 0.0;
        }
        if (RP_ABL == 3) {                                           // (timing only: G reads without bank conflicts)
#pragma unroll
            for (int s = 0; s < RP_SLOTS; ++s) tl[s] = (tid + 32 * s / RP_SLOTS * 0 + s * 37) % RP_TT;
        }
        const int clm = cl[0];                                       // (COLM: the slots' common column; a thread's first slot is never padding unless all are)
#ifdef RP_STAMP
        unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        // ---- walk 1: row totals
        constexpr int NSEG = CHK ? 8 : 1;
        const int seg_len = CHK ? (((K + RP_KC - 1) / RP_KC + NSEG - 1) / NSEG) * RP_KC : K;     // categories per eighth (whole chunks)
        double chk[NSEG][RP_SLOTS];
        fetch(0);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
#pragma unroll
            for (int s = 0; s < RP_SLOTS; ++s) chk[sg][s] = cum[s];
            const int qb = min(K, (sg + 1) * seg_len);
            for (int q0 = sg * seg_len; q0 < qb; q0 += RP_KC) {
                RP_ST_A();
                stage();
                __syncthreads();
                RP_ST_C();
                if (q0 + RP_KC < K) fetch(q0 + RP_KC);
#pragma unroll 4
                for (int kk = 0; kk < (RP_ABL == 1 ? 0 : RP_KC); ++kk) {
                    const double ec = Et[kk][clm];
#pragma unroll
                    for (int s = 0; s < RP_SLOTS; ++s) cum[s] = cum[s] + Gt[kk][tl[s]] * (COLM ? ec : Et[kk][cl[s]]);
                }
                RP_ST_D(0);
            }
        }
        // first thresholds; the baseline category
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s) {
            total[s] = cum[s];
            if (n[s] > 0) {
                const int c = c0 + cl[s];
                const uint64_t bin = (uint64_t)(t0 + tl[s]) + (uint64_t)T * (uint64_t)c;
                u[s] = rp_next_u(0.0, n[s], seed, step, bin, 0);
                thr[s] = u[s] * total[s];
                cum[s] = baseT ? baseT[(size_t)(t0 + tl[s]) + (size_t)T * c] : base[c];
                while (j[s] < n[s] && cum[s] > thr[s]) {
                    atomicAdd(&counts[c], 1);
                    if (base_counts) atomicAdd(&base_counts[(size_t)(t0 + tl[s]) + (size_t)T * c], 1);
                    if (++j[s] < n[s]) { u[s] = rp_next_u(u[s], n[s] - j[s], seed, step, bin, j[s]); thr[s] = u[s] * total[s]; }
                }
            }
            if (j[s] >= n[s]) thr[s] = __builtin_inf();                      // nothing (left) to place: walk 2 never stops here
        }
        // (CHK) where a bin's chain is entered: the last eighth whose starting sum has not passed the bin's next threshold
        // (sums of non-negative terms: the running sum never decreases, so nothing is placed before that point)
        int from[RP_SLOTS];
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s) {
            from[s] = 0;
            if (CHK) {
                int sg1 = 0;
                double start = chk[0][s];
#pragma unroll
                for (int sg = 1; sg < NSEG; ++sg) { const bool le = chk[sg][s] <= thr[s]; sg1 += le ? 1 : 0; start = le ? chk[sg][s] : start; }
                if (thr[s] < __builtin_inf()) { cum[s] = start; from[s] = sg1 * seg_len; }
            }
        }
        if (CHK && XCHG && nbins <= RP_TH * RP_SLOTS) {                      // (one round of bins: the list's LDS is free as well)
            static_assert(!XCHG || RP_SLOTS == 2, "the exchange hands a thread two consecutive places");
            constexpr int XCAP = RP_TH * RP_SLOTS, XSIDE = 256;
            static_assert(!XCHG || 256 + XCAP * 20 + XSIDE * 24 <= 8 * RP_KC * (RP_TT + RP_CT + 1) + 2 * RP_TT * RP_CT, "exchange buffers fit the kernel's LDS");
            int *xc = reinterpret_cast<int *>(rp_smem);                       // [0..7] bins per eighth, [8] bins with several events left, [9] gave up, [16..24] starts
            unsigned int *xpk = reinterpret_cast<unsigned int *>(rp_smem + 256);                  // [XCAP] tl | cl << 8 | eighth << 16 | several << 19 | side index << 20
            double *xthr = reinterpret_cast<double *>(rp_smem + 256 + 4 * XCAP);                   // [XCAP]
            double *xcum = xthr + XCAP;                                                            // [XCAP]
            double *xtot = xcum + XCAP, *xu = xtot + XSIDE;                                        // [XSIDE] each
            unsigned int *xnj = reinterpret_cast<unsigned int *>(xu + XSIDE);                      // [XSIDE] n | j << 16
            __syncthreads();                                                  // walk 1's chunk is done with
            if (tid < 32) xc[tid] = 0;
            __syncthreads();
            int rank[RP_SLOTS], side[RP_SLOTS];
#pragma unroll
            for (int s = 0; s < RP_SLOTS; ++s) {
                rank[s] = -1; side[s] = -1;
                if (thr[s] < __builtin_inf()) {
                    rank[s] = atomicAdd(&xc[from[s] / seg_len], 1);
                    if (n[s] - j[s] > 1) { side[s] = atomicAdd(&xc[8], 1); if (side[s] >= XSIDE || n[s] > 65535) xc[9] = 1; }
                }
            }
            __syncthreads();
            if (tid == 0) { int acc = 0; for (int g = 0; g < 8; ++g) { xc[16 + g] = acc; acc += xc[g]; } xc[24] = acc; }
            __syncthreads();
            if (xc[9] == 0) {                                                 // (uniform)
#pragma unroll
                for (int s = 0; s < RP_SLOTS; ++s) {
                    if (rank[s] >= 0) {
                        const int sg = from[s] / seg_len, d = xc[16 + sg] + rank[s];
                        xpk[d] = (unsigned int)tl[s] | (unsigned int)cl[s] << 8 | (unsigned int)sg << 16 | (side[s] >= 0 ? 1u << 19 | (unsigned int)side[s] << 20 : 0u);
                        xthr[d] = thr[s]; xcum[d] = cum[s];
                        if (side[s] >= 0) { xtot[side[s]] = total[s]; xu[side[s]] = u[s]; xnj[side[s]] = (unsigned int)n[s] | (unsigned int)j[s] << 16; }
                    }
                }
                __syncthreads();
                const int nact = xc[24];
#pragma unroll
                for (int s = 0; s < RP_SLOTS; ++s) {
                    const int d = tid * RP_SLOTS + s;
                    n[s] = 0; j[s] = 0; thr[s] = __builtin_inf(); from[s] = 0;
                    if (d < nact) {
                        const unsigned int pk = xpk[d];
                        tl[s] = (int)(pk & 255u); cl[s] = (int)(pk >> 8 & 255u); from[s] = (int)(pk >> 16 & 7u) * seg_len;
                        thr[s] = xthr[d]; cum[s] = xcum[d];
                        n[s] = 1;                                             // one event left: no further threshold, total and u unused
                        if (pk >> 19 & 1u) {
                            const int m = (int)(pk >> 20);
                            total[s] = xtot[m]; u[s] = xu[m]; n[s] = (int)(xnj[m] & 65535u); j[s] = (int)(xnj[m] >> 16);
                        }
                    }
                }
            }
        }
        // ---- walk 2: categories by inverse CDF
        fetch(0);
        for (int q0 = 0; q0 < K; q0 += RP_KC) {
            RP_ST_A();
            stage();
            __syncthreads();
            RP_ST_C();
            if (q0 + RP_KC < K) fetch(q0 + RP_KC);
#pragma unroll
            for (int s = 0; s < RP_SLOTS; ++s) {
              if (CHK && !(q0 >= from[s] && thr[s] < __builtin_inf())) continue;
              if (RP_ABL != 1) {
                  // The chunk's running sums in straight-line code, its LDS reads in flight together, every partial sum kept.  The
                  // sums never decrease, so a threshold passed inside the chunk is placed at the first category whose sum exceeds it:
                  // the number of partial sums at or below it -- compares, not a loop with an LDS round trip per category.  (That
                  // loop ran in nearly every chunk: 8 waves x 128 bins place an event in one chunk of 256 each, so some wave of the
                  // workgroup was always in it and the others waited at the barrier: 62 % of walk 2.)
                  if (!CHK) {                                                  // (every lane in every chunk: the partial sums would spill)
                      double e0 = cum[s];
#pragma unroll
                      for (int kk = 0; kk < RP_KC; ++kk) e0 = e0 + Gt[kk][tl[s]] * Et[kk][COLM ? clm : cl[s]];
                      if (!(e0 > thr[s])) { cum[s] = e0; continue; }
                      for (int kk = 0; kk < RP_KC; ++kk) {
                          cum[s] = cum[s] + Gt[kk][tl[s]] * Et[kk][COLM ? clm : cl[s]];
                          if (cum[s] > thr[s]) {
                              const int c = c0 + cl[s];
                              const uint64_t bin = (uint64_t)(t0 + tl[s]) + (uint64_t)T * (uint64_t)c;
                              do {
                                  atomicAdd(&counts[(size_t)c + (size_t)N * (1 + q0 + kk)], 1);
                                  if (++j[s] < n[s]) { u[s] = rp_next_u(u[s], n[s] - j[s], seed, step, bin, j[s]); thr[s] = u[s] * total[s]; }
                                  else thr[s] = __builtin_inf();
                              } while (cum[s] > thr[s]);
                          }
                      }
                      continue;
                  }
                  double pre[RP_KC];
                  double e = cum[s];
#pragma unroll
                  for (int kk = 0; kk < RP_KC; ++kk) { e = e + Gt[kk][tl[s]] * Et[kk][cl[s]]; pre[kk] = e; }
                  if (e > thr[s]) {
                      const int c = c0 + cl[s];
                      const uint64_t bin = (uint64_t)(t0 + tl[s]) + (uint64_t)T * (uint64_t)c;
                      do {
                          int kx = 0;
#pragma unroll
                          for (int kk = 0; kk < RP_KC; ++kk) kx += pre[kk] <= thr[s] ? 1 : 0;
                          atomicAdd(&counts[(size_t)c + (size_t)N * (1 + q0 + kx)], 1);
                          if (++j[s] < n[s]) { u[s] = rp_next_u(u[s], n[s] - j[s], seed, step, bin, j[s]); thr[s] = u[s] * total[s]; }
                          else thr[s] = __builtin_inf();
                      } while (e > thr[s]);
                  }
                  cum[s] = e;
              }
            }
            RP_ST_D(1);
        }
#ifdef RP_STAMP
        if (tid == 0) {
            const unsigned int wg = blockIdx.x + gridDim.x * blockIdx.y;
            if (wg < 4096) for (int k = 0; k < 8; ++k) g_rp_stamps[8 * wg + k] = st[k];
        }
#endif
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s)                                    // capped at the last category
            if (j[s] < n[s]) atomicAdd(&counts[(size_t)(c0 + cl[s]) + (size_t)N * K], n[s] - j[s]);
    }
}

// Runs the parent-count sweep; *d_counts_out (int [N*(1+K)], index c + N*k) stays in ctx scratch until the next call.
static nhp_status disc_parent_counts(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0, const double *W,
                                     const double *theta, const double *A, double dt, uint64_t seed, uint64_t step,
                                     size_t extra_doubles, int **d_counts_out, double **extra_out, double **dW_out,
                                     double **dth_out, double **dl0_out)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, K = N * (size_t)ds->B, NC = N * (1 + K);
    double *E2, *base, *extra;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E2, &base, (NC + 1) / 2 + 1 + extra_doubles, &extra, 1));
    int *d_counts = reinterpret_cast<int *>(extra);
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int) * NC, st));
    if (ds->T >= ((int64_t)1 << 31) || K >= ((size_t)1 << 24)) { nhp_set_error(ctx, "resample_parents: T or N*B too large"); return NHP_ENOTIMPL; }
    // q / B for q < 2^24 as a multiply-high: exact with magic = floor(2^32 / B) + 1 while q·B < 2^32
    const unsigned b_magic = (unsigned)((((uint64_t)1 << 32) / (uint64_t)ds->B + 1) & 0xFFFFFFFFu);   // unused for B = 1
    if (ds->d_base_counts) NHP_HIP(ctx, hipMemsetAsync(ds->d_base_counts, 0, sizeof(int) * (size_t)ds->T * N, st));
    // tile (bins x nodes, threads): the larger one halves the staging traffic per occupied bin where the problem fills it;
    // NHP_RP_TILE = "TT,CT,THREADS" overrides (64,128,256 | 128,128,512 | 128,256,1024), NHP_RP_SLOTS the slots per thread
    int TT = 64, CT = 128, TH = 256;
    if (N >= 256 && ds->T >= 128 * 256) { TT = 128; CT = 128; TH = 512; }      // (10.6 ms; 128 x 256 x 1024 11.9, 64 x 128 x 256 12.4)
    if (const char *ts = getenv("NHP_RP_TILE")) sscanf(ts, "%d,%d,%d", &TT, &CT, &TH);
    if (!((TT == 64 && CT == 128 && TH == 256) || (TT == 128 && CT == 128 && TH == 512) || (TT == 128 && CT == 256 && TH == 1024))) { TT = 64; CT = 128; TH = 256; }
    const int ntx = (int)((ds->T + TT - 1) / TT), ncy = (int)((N + CT - 1) / CT);
    const int xcd_env = getenv("NHP_RP_XCD") ? atoi(getenv("NHP_RP_XCD")) : 1;
    const int xcd_ncy = xcd_env && ncy > 1 ? ncy : 0;
    dim3 grid(xcd_ncy ? (unsigned)(((ntx + 7) / 8) * 8 * ncy) : (unsigned)ntx, xcd_ncy ? 1u : (unsigned)ncy);
    // occupied bins of a tile: the mean plus three standard deviations (a tile that overflows its slots walks twice)
    const double mean = (double)ds->nocc * (double)(TT * CT) / ((double)ds->T * (double)std::max<size_t>(N, (size_t)CT));
    // column-major list (slots of a thread share their node): every column of the tile is padded to a multiple of the slots
    const int colm_env = getenv("NHP_RP_COLM") ? atoi(getenv("NHP_RP_COLM")) : 1;
    const bool colm = colm_env != 0 && TT >= 64 && (TT * CT / (TH / 64)) % TT == 0;
    const char *fs = getenv("NHP_RP_SLOTS");
    auto need_for = [&](int sl) { return mean + 3.0 * sqrt(mean) + (colm ? 0.5 * (sl - 1) * CT : 0.0); };
    // (four bins a thread measured 17-35 ms against 9-12 with two and a second round over the tile's list: only by request)
    const int slots = fs ? atoi(fs) : (need_for(1) <= 1.0 * TH ? 1 : 2);
    const size_t lds = 8 * (size_t)RP_KC * (size_t)(TT + CT + 1) + 2 * (size_t)TT * CT;      // (a list of 2048 entries instead: no more workgroups per CU -- 128 registers)
#define RP_LAUNCH(S, tt, ct, th, cm, ck, xg)                                                                                      \
    do {                                                                                                                          \
        if (lds > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_disc_resample_parents<S, tt, ct, th, cm, ck, xg>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_disc_resample_parents<S, tt, ct, th, cm, ck, xg>), grid, dim3(th), lds, st, ds->d_dataT, ds->d_conv, E2, base, \
                           lambda0 ? nullptr : ds->d_baseT, ds->T, ds->N, ds->B, b_magic, seed, step, d_counts, ds->d_base_counts, xcd_ncy); \
    } while (0)
#define RP_SL(tt, ct, th, cm, ck) do { if (slots == 1) RP_LAUNCH(1, tt, ct, th, cm, ck, false); else if (slots == 2) { if (ck && xchg) RP_LAUNCH(2, tt, ct, th, cm, ck, ck); else RP_LAUNCH(2, tt, ct, th, cm, ck, false); } else RP_LAUNCH(4, tt, ct, th, cm, ck, false); } while (0)
#define RP_CK(tt, ct, th, cm) do { if (chk) RP_SL(tt, ct, th, cm, true); else RP_SL(tt, ct, th, cm, false); } while (0)
#define RP_TILE(tt, ct, th) do { if (colm) RP_CK(tt, ct, th, true); else RP_CK(tt, ct, th, false); } while (0)
    const int chk_env = getenv("NHP_RP_CHK") ? atoi(getenv("NHP_RP_CHK")) : 1;
    const bool chk = chk_env != 0;
    const bool xchg = !(getenv("NHP_RP_XCHG") && atoi(getenv("NHP_RP_XCHG")) == 0);
    if (TT == 64) RP_TILE(64, 128, 256); else if (CT == 128) RP_TILE(128, 128, 512); else RP_TILE(128, 256, 1024);
#undef RP_CK
#undef RP_TILE
#undef RP_SL
#undef RP_LAUNCH
    if (ds->d_base_counts) const_cast<nhp_disc_dataset *>(ds)->base_counts_valid = true;
    NHP_HIP(ctx, hipGetLastError());
    *d_counts_out = d_counts;
    if (extra_out) *extra_out = extra + (NC + 1) / 2 + 1;
    // stage_bump's layout after E: base (N) | λ0 (N) | W (N²) | θ (N²B) | A (N²) | extra
    if (dl0_out) *dl0_out = base + N;
    if (dW_out) *dW_out = base + 2 * N;
    if (dth_out) *dth_out = base + 2 * N + NN;
    return NHP_OK;
}

extern "C" nhp_status nhp_disc_resample_parents(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                                const double *W, const double *theta, const double *A, double dt,
                                                uint64_t seed, uint64_t step, int64_t *counts)
{
    if (!ctx || !ds || !counts) return NHP_EINVAL;
    const size_t N = (size_t)ds->N, NC = N * (1 + N * (size_t)ds->B);
    int *d_counts;
    NHP_TRY(disc_parent_counts(ctx, ds, lambda0, W, theta, A, dt, seed, step, 0, &d_counts, nullptr, nullptr, nullptr, nullptr));
    hipStream_t st = ctx->stream;
    std::vector<int> h((size_t)NC);
    NHP_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, sizeof(int) * NC, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < NC; ++i) counts[i] = h[i];
    return NHP_OK;
}

// ---- conjugate draws of the discrete resample! methods on the device (src/baselines.jl:413-419 as intended (D2),
// src/weights.jl:59-64, src/impulses.jl:337-353): λ0_c ~ Gamma(α0 + counts[c, 0], 1/(β0 + T·dt));
// W[p,c] ~ Gamma(κ + Σ_b counts[c, 1+pB+b], 1/(ν + Σ_t data[p, t])); θ[p,c,:] ~ Dirichlet(γ + counts[c, 1+pB+·])
// as normalised Gamma(·, 1) draws.  Philox-keyed like the continuous draws: distributional parity with Julia.
__global__ __launch_bounds__(256) void k_disc_gibbs_draw(int N, int B, double Tdt, double alpha0, double beta0, double kappa,
                                                         double nu, double gamma0, uint64_t seed, uint64_t step,
                                                         const int *__restrict__ counts, const double *__restrict__ node_counts,
                                                         double *__restrict__ lambda0, double *__restrict__ W,
                                                         double *__restrict__ theta)
{
    const size_t NN = (size_t)N * N;
    const size_t pc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pc < (size_t)N)
        lambda0[pc] = dev_gamma(alpha0 + (double)counts[pc], 1.0 / (beta0 + Tdt), seed ^ 0x243F6A8885A308D3ull, step, pc);
    if (pc >= NN) return;
    const size_t p = pc % N, c = pc / N;
    double m = 0.0, gs = 0.0;
    for (int b = 0; b < B; ++b) {
        const double cnt = (double)counts[c + (size_t)N * (1 + p * B + b)];
        m += cnt;
        const double gv = dev_gamma(gamma0 + cnt, 1.0, seed ^ 0xA4093822299F31D0ull, step, pc + (size_t)b * NN);
        theta[pc + (size_t)b * NN] = gv;
        gs += gv;
    }
    for (int b = 0; b < B; ++b) theta[pc + (size_t)b * NN] /= gs;
    W[pc] = dev_gamma(kappa + m, 1.0 / (nu + node_counts[p]), seed ^ 0x13198A2E03707344ull, step, pc);
}

// resample!(process::DiscreteStandardHawkesProcess, data, convolved) src/discrete.jl:362-368 in one call: parent
// counts, then the conjugate draws, all on the device; lambda0 [N], W [N*N], theta [N*N*B] are read and overwritten.
extern "C" nhp_status nhp_disc_gibbs_step(nhp_ctx *ctx, const nhp_disc_dataset *ds, double *lambda0, double *W,
                                          double *theta, const double *A, double dt, double alpha0, double beta0,
                                          double kappa, double nu, double gamma0, uint64_t seed, uint64_t step)
{
    if (!ctx || !ds || !lambda0 || !W || !theta) return NHP_EINVAL;
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B;
    int *d_counts;
    double *dW, *dth, *dl0;
    NHP_TRY(disc_parent_counts(ctx, ds, lambda0, W, theta, A, dt, seed, step, 0, &d_counts, nullptr, &dW, &dth, &dl0));
    hipStream_t st = ctx->stream;
    // Σ_t data[p, t] = node_counts(data): the first half of the dataset's column statistics
    hipLaunchKernelGGL(k_disc_gibbs_draw, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, (double)ds->T * dt,
                       alpha0, beta0, kappa, nu, gamma0, seed, step, d_counts, ds->d_colsum, dl0, dW, dth);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(lambda0, dl0, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(W, dW, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(theta, dth, 8 * NN * B, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

// ---- discrete adjacency Gibbs sweep (reference resample_adjacency_matrix! / resample_column!
// / conditional_loglikelihood src/discrete.jl:424-480).  For entry (p, c) the reference evaluates two full
// Poisson log-likelihoods of column c over all T bins and all N·B parent terms.  Their difference is
//     ll1 - ll0 = Σ_{t: s>0} s_tc [log(λ⁰_tc + x_t) - log λ⁰_tc] - Σ_t x_t + log ρ - log(1-ρ),
//     x_t = W[p,c] dt Σ_b Ŝ[t,p,b] θ[p,c,b],   λ⁰ = the intensity with A[p,c] = 0,
// where only OCCUPIED bins need a log and Σ_t x_t = Σ_b (W θ dt)[p,c,b] · Σ_t Ŝ[t,p,b] uses per-dataset column
// sums.  Columns are independent, entries of a column sequential in p -- so the sweep is N steps, each over
// all occupied bins of all columns at once, ONE launch per step (k_dadj_step).  λ of the occupied bins is carried
// incrementally.
//
// Round 3 (PMC pass on the two-kernel step of round 2, profiles/r03_pmc/dadj_accum_counters.txt: three waves per SIMD, each
// waiting 80 % of its life on a chain of dependent loads -- entry -> dprev[c], V[c,·] out of L2 at ~220 cycles a request ->
// arithmetic -- 13 entries a thread one after the other; then 16.5 µs of a 16-workgroup reduction kernel):
//  * what does not depend on the decisions is tabulated once per sweep (k_dadj_tables): V[p][c][·] = W θ dt row-major in p,
//    A's rows, logit(u), the two logarithms of ρ and Σ_t x_t;
//  * the time axis is cut into spans of at most 256 bins when the dataset is made (a balanced round of two workgroups per CU),
//    a span's occupied bins sorted by (node, bin): a thread takes consecutive entries -- 16-byte reads of the packed words
//    (node | count | bin % 256) and of λ, requested while the tables are staged -- that mostly share their column, so row c
//    of V, a_p[c] and the flip of (p - 1, c) are read once per run and the run's sum reaches the column's LDS accumulator
//    as one atomic;
//  * x of the previous step is recomputed for the columns that flipped instead of being stored and re-read for all;
//  * log((l0 + x)/l0) is an atanh series where x is small against l0 (dadj_logratio);
//  * a workgroup leaves its row of partial sums, the last of each group of NHP_DA_GROUP adds its group's rows (every row
//    requested before the first is used: a serial `sum += load` is one L2 round trip per row) and adds the result, with
//    atomics, into the step's ONE row of sums (three rows in rotation, the one a launch adds to zeroed two launches before);
//    row p is DECIDED AT THE START OF LAUNCH p + 1 by every workgroup for itself from that row -- a second ticket and one
//    deciding workgroup at the end of the launch, with everybody else gone, cost 3.5 µs more a step; 32 group rows read by
//    every workgroup (128 KB each, 64 MB a step out of L2) 2 µs more.  The order of a column's 32 atomic additions varies
//    from run to run, as does the order of the LDS atomics inside a workgroup: two runs differ where |d - logit u| is
//    within rounding of zero.
// 55 µs (two launches) -> 27 µs a step at config-4 scale; 36.2 -> 20.0 ms per sweep, 4.1 ms of which are k_dadj_lambda0
// before the first step (the T x N intensity GEMM and a gather before: 6.3 ms).
__global__ __launch_bounds__(256) void k_dadj_gather(const double *__restrict__ lam, const int32_t *__restrict__ occ_t,
                                                     const int32_t *__restrict__ occ_c, int64_t nocc, int64_t T,
                                                     double *__restrict__ lam_occ)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nocc) lam_occ[i] = lam[(size_t)occ_t[i] + (size_t)T * occ_c[i]];
}

// Vall[p][c*B + b] = W[p,c] θ[p,c,b] dt,  AT[p][c] = A[p,c],  LU[p][c] = logit(u[p,c]),  LR1 / LR2 = log ρ, log(1 - ρ),
// SX[p][c] = Σ_t x_t
__global__ __launch_bounds__(256) void k_dadj_tables(int N, int B, double dt, const double *__restrict__ W, const double *__restrict__ theta,
                                                     const double *__restrict__ A, const double *__restrict__ rho_mat, double rho_scalar,
                                                     const double *__restrict__ u, uint64_t seed, uint64_t step, double *__restrict__ Vall,
                                                     double *__restrict__ AT, double *__restrict__ LU, double *__restrict__ LR1,
                                                     double *__restrict__ LR2, const double *__restrict__ convsum, double *__restrict__ SX)
{
    // a 16 x 16 patch of (p, c) per workgroup: reads run along p (the tables are column-major), writes along c
    __shared__ double tw[16][17], ta[16][17], tu[16][17], t1[16][17], t2[16][17];
    const int lp = threadIdx.x & 15, lc = threadIdx.x >> 4;
    const int p0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
    const size_t NN = (size_t)N * N;
    {
        const int p = p0 + lp, c = c0 + lc;
        if (p < N && c < N) {
            const size_t pc = (size_t)p + (size_t)c * N;
            const double rho = rho_mat ? rho_mat[pc] : rho_scalar;
            const double uu = u ? u[pc] : nhp_philox_uniform(seed ^ 0xAD7AC3117D15C0DEull, step, pc);
            tw[lc][lp] = W[pc]; ta[lc][lp] = A[pc];
            tu[lc][lp] = nhp_log(uu / (1.0 - uu)); t1[lc][lp] = nhp_log(rho); t2[lc][lp] = nhp_log(1.0 - rho);
        }
    }
    __syncthreads();
    const int wp = threadIdx.x >> 4, wc = threadIdx.x & 15;      // now lanes run along c
    const int p = p0 + wp, c = c0 + wc;
    if (p >= N || c >= N) return;
    const size_t row = (size_t)p * N + c, pc = (size_t)p + (size_t)c * N;
    AT[row] = ta[wc][wp]; LU[row] = tu[wc][wp]; LR1[row] = t1[wc][wp]; LR2[row] = t2[wc][wp];
    const double w = tw[wc][wp];
    double sx = 0.0;                                                  // Σ_t x_t = Σ_b (W θ dt)[p,c,b] · Σ_t Ŝ[t,p,b]
    for (int b = 0; b < B; ++b) {
        const double vb = (w * theta[pc + (size_t)b * NN]) * dt;
        Vall[row * B + b] = vb;
        sx += vb * convsum[(size_t)p + (size_t)N * b];
    }
    SX[row] = sx;
}

#ifndef NHP_DA_GROUP
#define NHP_DA_GROUP 16     // workgroups whose rows one of them adds; the next launch reads nspans / GROUP rows (4: 23.1, 8: 22.6, 16: 22.5, 32: 24.2 ms per sweep)
#endif
#ifndef DADJ_ABL
#define DADJ_ABL 0      // (timing ablations: wrong results)
#endif
template <int BT>
__device__ __forceinline__ double dadj_x(const double *__restrict__ G, int tt, const double *__restrict__ v, int B)
{
    double x = 0.0;
    if (BT > 0) {
#pragma unroll
        for (int b = 0; b < BT; ++b) x += G[b * NHP_DA_SPAN + tt] * v[b];
    } else {
        for (int b = 0; b < B; ++b) x += G[b * NHP_DA_SPAN + tt] * v[b];
    }
    return x;
}

// log((l0 + x) / l0) for the occupied bin of one entry.  One parent's share x of a bin's intensity is small against the rest
// l0 almost always: then 2 atanh(s), s = x / (2 l0 + x), as its odd series (|s| < 1/8: the first omitted term is 2e-16 of the
// value) with the quotient from the hardware reciprocal and two Newton steps -- a quarter of the instructions of two
// logarithms, and without their cancellation.  Elsewhere the difference of the two logarithms as the reference writes it.
__device__ __attribute__((noinline)) double dadj_logdiff(double l0, double x) { return nhp_log(l0 + x) - nhp_log(l0); }

__device__ __forceinline__ double dadj_logratio(double l0, double x)
{
    const double den = 2.0 * l0 + x;
    double r = __builtin_amdgcn_rcp(den);
    r = r * (2.0 - den * r);
    r = r * (2.0 - den * r);
    const double sq = x * r;
    if (!(sq < 0.125) || !(den > 0.0)) return dadj_logdiff(l0, x);       // (a call: twelve inlined copies of two logarithms are 30 KB of code)
    const double z = sq * sq;
    const double poly = 2.0 + z * (2.0 / 3.0 + z * (2.0 / 5.0 + z * (2.0 / 7.0 + z * (2.0 / 9.0 + z * (2.0 / 11.0 + z * (2.0 / 13.0 + z * (2.0 / 15.0 + z * (2.0 / 17.0))))))));
    return sq * poly;
}

struct nhp_dadj_args {
    int N, B, nspans;
    int gsz;                             // doubles of the LDS region that holds the two Ŝ spans, then the decision's partial sums
    int64_t T;
    const double *conv;
    const uint32_t *occ_pack;
    const int32_t *occ_t, *occ_c, *occ_off, *span_t;     // workgroup k: entries [occ_off[k], occ_off[k+1]), bins [span_t[k], span_t[k+1])
    const double *occ_s;
    const double *Vall, *AT, *LU, *LR1, *LR2, *SX;
    double *lam_occ;
    double *partial, *gpartial;          // [workgroups][N], [3][N]: step p's sums in row p mod 3
    unsigned int *tick;                  // [groups] words 32 apart
    double *A;
    unsigned long long *stamps;          // (DADJ_STAMP builds: 8 per workgroup)
};

// λ of the occupied bins under the current adjacency matrix -- the sweep's starting point -- without the T x N intensity GEMM
// (6.2 ms at config-4 scale for the 5 % of bins that are occupied).  The spans and entry lists of the step kernel: a workgroup
// keeps its entries' λ in registers and walks the N parents once, row p of V and of A and its span of Ŝ staged per parent
// (requested one parent ahead): λ += a[p,c] · Σ_b Ŝ[t,p,b] V[p,c,b], parents in order, b innermost -- the order of the
// reference's own sum.  No exchange between workgroups, one launch.  Entries beyond 12 a thread: the GEMM.
template <int BT, int TH>
__global__ __launch_bounds__(TH, TH / 128) void k_dadj_lambda0(nhp_dadj_args a, const double *__restrict__ base, const double *__restrict__ baseT)
{
    extern __shared__ __align__(16) double dsm[];
    constexpr int B = BT, PRE = 3;
    const int N = a.N, tid = threadIdx.x;
    double *Gt = dsm;                                       // [B][SPAN]
    double *apl = Gt + (size_t)B * NHP_DA_SPAN;             // [N]
    double *Vl = apl + N;                                   // [N·B]
    const int64_t t0 = a.span_t[blockIdx.x];
    const int span = a.span_t[blockIdx.x + 1] - (int)t0;
    const int i0 = a.occ_off[blockIdx.x], i1 = a.occ_off[blockIdx.x + 1];
    const int per = ((i1 - i0 + 4 * TH - 1) / (4 * TH)) * 4;            // <= 4·PRE (host)
    const int mine = i0 + tid * per, mend = min(mine + per, i1);
    const int tb = (int)(t0 % NHP_DA_SPAN);
    uint4 pw[PRE];
    double lam[4 * PRE];
#pragma unroll
    for (int g = 0; g < PRE; ++g) {
        const int bs = mine + 4 * g;
        pw[g] = make_uint4(0u, 0u, 0u, 0u);
        if (bs < mend) pw[g] = *reinterpret_cast<const uint4 *>(a.occ_pack + bs);
        const uint32_t w4[4] = {pw[g].x, pw[g].y, pw[g].z, pw[g].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (int)(w4[j] >> 16), tt = ((int)(w4[j] & 255u) - tb) & (NHP_DA_SPAN - 1);
            lam[4 * g + j] = bs < mend ? (baseT ? baseT[(size_t)(t0 + tt) + (size_t)a.T * c] : base[c]) : 0.0;
        }
    }
    constexpr int GN = (B * NHP_DA_SPAN + TH - 1) / TH, VN = 4;
    const int nv2 = N * B / 2;
    double rg[GN];
    double2 rv[VN];
    double ra[2];
    auto fetch = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < GN; ++r) {
            const int e = tid + TH * r, b = e / NHP_DA_SPAN, tt = e % NHP_DA_SPAN;
            const int64_t t = t0 + tt;
            const bool ok = e < B * NHP_DA_SPAN && tt < span && t < a.T;
            rg[r] = ok ? a.conv[(size_t)t + (size_t)a.T * ((size_t)p + (size_t)N * b)] : 0.0;
        }
        const double2 *src = reinterpret_cast<const double2 *>(a.Vall + (size_t)p * N * B);
#pragma unroll
        for (int r = 0; r < VN; ++r) { const int e = tid + TH * r; rv[r] = e < nv2 ? src[e] : make_double2(0.0, 0.0); }
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int c = tid + TH * r; ra[r] = c < N ? a.AT[(size_t)p * N + c] : 0.0; }
    };
    fetch(0);
    for (int p = 0; p < N; ++p) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GN; ++r) { const int e = tid + TH * r; if (e < B * NHP_DA_SPAN) Gt[e] = rg[r]; }
        {
            double2 *dst = reinterpret_cast<double2 *>(Vl);
#pragma unroll
            for (int r = 0; r < VN; ++r) { const int e = tid + TH * r; if (e < nv2) dst[e] = rv[r]; }
            for (int e = tid + TH * VN; e < nv2; e += TH) dst[e] = reinterpret_cast<const double2 *>(a.Vall + (size_t)p * N * B)[e];
#pragma unroll
            for (int r = 0; r < 2; ++r) { const int c = tid + TH * r; if (c < N) apl[c] = ra[r]; }
            for (int c = tid + 2 * TH; c < N; c += TH) apl[c] = a.AT[(size_t)p * N + c];
        }
        __syncthreads();
        if (p + 1 < N) fetch(p + 1);
        int c_cur = -1;
        double apc = 0.0;
        double v[B];
#pragma unroll
        for (int g = 0; g < PRE; ++g) {
            if (mine + 4 * g < mend) {
                const uint32_t w4[4] = {pw[g].x, pw[g].y, pw[g].z, pw[g].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (int)(w4[j] >> 16);
                    if (c != c_cur) {
                        c_cur = c; apc = apl[c];
                        if (apc != 0.0) {
#pragma unroll
                            for (int b = 0; b < B; ++b) v[b] = Vl[(size_t)c * B + b];
                        }
                    }
                    if (apc != 0.0) {
                        const int tt = ((int)(w4[j] & 255u) - tb) & (NHP_DA_SPAN - 1);
                        lam[4 * g + j] += apc * dadj_x<BT>(Gt, tt, v, B);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < PRE; ++g) {
        const int bs = mine + 4 * g;
        if (bs < mend) {
            *reinterpret_cast<double2 *>(a.lam_occ + bs) = make_double2(lam[4 * g], lam[4 * g + 1]);
            *reinterpret_cast<double2 *>(a.lam_occ + bs + 2) = make_double2(lam[4 * g + 2], lam[4 * g + 3]);
        }
    }
}

#ifdef DADJ_STAMP
#define DADJ_ST(k) do { if (tid == 0 && a.stamps) a.stamps[8 * (size_t)blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DADJ_ST(k) do { } while (0)
#endif
template <int BT, bool PACK, bool VLDS, int TH>
__global__ __launch_bounds__(TH, TH / 128) void k_dadj_step(int p, nhp_dadj_args a)
{
    extern __shared__ __align__(16) double dsm[];
    const int N = a.N, B = BT > 0 ? BT : a.B, tid = threadIdx.x;
    double *Gt = dsm;                                       // [B][SPAN]
    double *Gp = Gt + (size_t)B * NHP_DA_SPAN;              // [B][SPAN]  (step p - 1)
    double *acc = dsm + a.gsz;                              // [N]
    double *apl = acc + N, *dpl = apl + N;                  // [N] each
    double *Vl = dpl + N;                                   // [N·B] (VLDS)
    __shared__ int flag;
    DADJ_ST(0);
    const int64_t t0 = a.span_t[blockIdx.x];
    const int span = a.span_t[blockIdx.x + 1] - (int)t0;    // <= SPAN
    const double *Vc = a.Vall + (size_t)p * N * B, *Vp = a.Vall + (size_t)(p > 0 ? p - 1 : 0) * N * B;
    // A thread takes `per` consecutive entries (a multiple of 4: 16-byte reads of the entry words, 2 x 16 bytes of λ).  Its first
    // DADJ_PRE groups of four are requested while the tables are staged.
    const int i0 = a.occ_off[blockIdx.x], i1 = a.occ_off[blockIdx.x + 1];          // (multiples of 4: the spans are padded)
    const int per = ((i1 - i0 + 4 * TH - 1) / (4 * TH)) * 4;
    const int mine = i0 + tid * per, mend = DADJ_ABL == 1 ? mine : min(mine + per, i1);
    constexpr int DADJ_PRE = 3;
    uint4 pre_w[DADJ_PRE];
    double2 pre_a[DADJ_PRE], pre_b[DADJ_PRE];
    // ---- row p - 1 is decided here, by every workgroup for itself (the group rows of step p - 1 are 128 KB out of L2: cheaper
    //      than a second ticket and a deciding workgroup at the end of that launch with everybody else gone); workgroup 0
    //      writes A.  Launch p = N does only this.
    const unsigned int nwg = a.nspans, grp = blockIdx.x / NHP_DA_GROUP, ngrp = (nwg + NHP_DA_GROUP - 1) / NHP_DA_GROUP;
    const unsigned int gfirst = grp * NHP_DA_GROUP, gsize = min((unsigned int)NHP_DA_GROUP, nwg - gfirst);
    auto decide = [&]() __attribute__((always_inline)) {
        // (the group rows of step p - 1 were added into ONE row by their groups' last workgroups: 4 KB to read here, not 128)
        const double *gp = a.gpartial + (size_t)((p - 1) % 3) * N;
        if (blockIdx.x == 0 && p < N) for (int c = tid; c < N; c += TH) a.gpartial[(size_t)((p + 1) % 3) * N + c] = 0.0;   // for the launch after this one
        for (int c = tid; c < N; c += TH) {
            const size_t row = (size_t)(p - 1) * N + c;
            const double lu = a.LU[row], l1 = a.LR1[row], l2 = a.LR2[row], sx = a.SX[row], aold = a.AT[row];
            const double delta = gp[c];
            const double d = (delta - sx) + l1 - l2;                                     // ll1 - ll0
            // rand(Bernoulli(exp(ll1 - logsumexp(ll0, ll1)))): u <= 1/(1 + e^{-d})  <=>  logit(u) <= d
            const double anew = lu <= d ? 1.0 : 0.0;
            dpl[c] = anew - aold;
            if (blockIdx.x == 0) a.A[(size_t)(p - 1) + (size_t)c * N] = anew;
        }
    };
    if (p == N) { decide(); return; }
    // staging: every global load is requested before the first LDS store (one round trip, not one per table)
    constexpr int GN = (8 * NHP_DA_SPAN + TH - 1) / TH;        // Ŝ elements a thread stages when B <= 8 (more: the loop below)
    const bool flips = p > 0;                                   // (the span of p - 1 is staged whether or not a column flipped)
    {
        double rg[GN], rgp[GN], ra[2];
#pragma unroll
        for (int r = 0; r < GN; ++r) {
            const int e = tid + TH * r, b = e / NHP_DA_SPAN, tt = e % NHP_DA_SPAN;
            const int64_t t = t0 + tt;
            const bool ok = e < B * NHP_DA_SPAN && tt < span && t < a.T;
            rg[r] = ok ? a.conv[(size_t)t + (size_t)a.T * ((size_t)p + (size_t)N * b)] : 0.0;
            rgp[r] = ok && flips ? a.conv[(size_t)t + (size_t)a.T * ((size_t)(p - 1) + (size_t)N * b)] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = tid + TH * r;
            ra[r] = c < N ? a.AT[(size_t)p * N + c] : 0.0;
        }
        if (VLDS && DADJ_ABL != 4) {
            if ((N * B) % 2 == 0) {
                const double2 *src = reinterpret_cast<const double2 *>(Vc);
                double2 *dst = reinterpret_cast<double2 *>(Vl);
                for (int e = tid; e < N * B / 2; e += TH) dst[e] = src[e];
            } else {
                for (int e = tid; e < N * B; e += TH) Vl[e] = Vc[e];
            }
        }
        // (the entries' requests go out behind the tables': they return while the tables are written to LDS)
        if (PACK) {
#pragma unroll
            for (int g = 0; g < DADJ_PRE; ++g) {
                const int base = mine + 4 * g;
                if (base < mend) {
                    pre_w[g] = *reinterpret_cast<const uint4 *>(a.occ_pack + base);
                    pre_a[g] = *reinterpret_cast<const double2 *>(a.lam_occ + base);
                    pre_b[g] = *reinterpret_cast<const double2 *>(a.lam_occ + base + 2);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GN; ++r) { const int e = tid + TH * r; if (e < B * NHP_DA_SPAN) { Gt[e] = rg[r]; if (flips) Gp[e] = rgp[r]; } }
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int c = tid + TH * r; if (c < N) { acc[c] = 0.0; apl[c] = ra[r]; if (p == 0) dpl[c] = 0.0; } }
        for (int e = tid + TH * GN; e < B * NHP_DA_SPAN; e += TH) {                       // B > 8
            const int b = e / NHP_DA_SPAN, tt = e % NHP_DA_SPAN;
            const int64_t t = t0 + tt;
            Gt[e] = tt < span && t < a.T ? a.conv[(size_t)t + (size_t)a.T * ((size_t)p + (size_t)N * b)] : 0.0;
        }
        for (int c = tid + 2 * TH; c < N; c += TH) { acc[c] = 0.0; apl[c] = a.AT[(size_t)p * N + c]; if (p == 0) dpl[c] = 0.0; }
    }
    if (p > 0) decide();                        // (its requests queue behind the tables' and the entries': one round trip for all)
    if (flips)
        for (int e = tid + TH * GN; e < B * NHP_DA_SPAN; e += TH) {                           // B > 8
            const int b = e / NHP_DA_SPAN, tt = e % NHP_DA_SPAN;
            const int64_t t = t0 + tt;
            Gp[e] = tt < span && t < a.T ? a.conv[(size_t)t + (size_t)a.T * ((size_t)(p - 1) + (size_t)N * b)] : 0.0;
        }
    __syncthreads();
    DADJ_ST(1);
    const double *Vx = VLDS ? Vl : Vc;
    const int tb = (int)(t0 % NHP_DA_SPAN);
    // The entries are sorted by node, so a thread's entries mostly share their column: row c of V, a_p[c], dprev[c] are read
    // when the node changes, the sum stays in a register and reaches acc[c] -- one LDS atomic -- at the end of the run.
    int c_cur = -1;
    double run = 0.0, apc = 0.0, dpc = 0.0;
    double v[BT > 0 ? BT : 1], vp[BT > 0 ? BT : 1];
    const double *vrow = Vx, *vprow = Vp;
    auto entry = [&](int idx, int c, int tt, double sv, double lam) __attribute__((always_inline)) {
        if (c != c_cur) {
            if (c_cur >= 0 && run != 0.0) atomicAdd(&acc[c_cur], run);
            c_cur = c; run = 0.0; apc = apl[c]; dpc = DADJ_ABL == 7 ? 0.0 : dpl[c];
            vrow = Vx + (size_t)c * B; vprow = Vp + (size_t)c * B;
            if (BT > 0) {
#pragma unroll
                for (int b = 0; b < BT; ++b) v[b] = vrow[b];
                if (dpc != 0.0) {
#pragma unroll
                    for (int b = 0; b < BT; ++b) vp[b] = vprow[b];
                }
            }
        }
        if (dpc != 0.0) {                                              // entry (p-1, c) flipped: carry it into λ
            lam += dpc * (BT > 0 ? dadj_x<BT>(Gp, tt, vp, B) : dadj_x<BT>(Gp, tt, vprow, B));
            a.lam_occ[idx] = lam;
        }
        const double x = BT > 0 ? dadj_x<BT>(Gt, tt, v, B) : dadj_x<BT>(Gt, tt, vrow, B);
        if (x > 0.0) {
            const double l0 = lam - apc * x;
            if (DADJ_ABL == 2) run += sv * l0;
            else if (DADJ_ABL == 6) run += sv * (nhp_log(l0 + x) - nhp_log(l0));
            else run += sv * dadj_logratio(l0, x);
        }
    };
    auto packed4 = [&](int base, const uint4 q, const double2 la, const double2 lb) __attribute__((always_inline)) {
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
        const double l4[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            entry(base + j, (int)(w4[j] >> 16), ((int)(w4[j] & 255u) - tb) & (NHP_DA_SPAN - 1), (double)((w4[j] >> 8) & 255u), l4[j]);
    };
    if (PACK) {
#pragma unroll
        for (int g = 0; g < DADJ_PRE; ++g) {
            const int base = mine + 4 * g;
            if (base < mend) packed4(base, pre_w[g], pre_a[g], pre_b[g]);
        }
        for (int base = mine + 4 * DADJ_PRE; base < mend; base += 4)
            packed4(base, *reinterpret_cast<const uint4 *>(a.occ_pack + base), *reinterpret_cast<const double2 *>(a.lam_occ + base),
                    *reinterpret_cast<const double2 *>(a.lam_occ + base + 2));
    } else {
        for (int i = mine; i < mend; ++i) entry(i, a.occ_c[i], a.occ_t[i] - (int)t0, a.occ_s[i], a.lam_occ[i]);
    }
    if (c_cur >= 0 && run != 0.0) atomicAdd(&acc[c_cur], run);
    __syncthreads();
    DADJ_ST(2);
    // ---- tail: this workgroup's row; the last workgroup of each group adds its group's rows (fixed order; every row requested
    //      before the first is used: a serial `sum += load` is one L2 round trip per row)
    for (int c = tid; c < N; c += TH)
        __hip_atomic_store(&a.partial[(size_t)blockIdx.x * N + c], acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DADJ_ST(3);
    if (tid == 0) flag = __hip_atomic_fetch_add(&a.tick[32 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
    __syncthreads();
    DADJ_ST(4);
    if (!flag) return;
    double *gout = a.gpartial + (size_t)(p % 3) * N;
    for (int c = tid; c < N; c += TH) {
        double sum = 0.0;
        for (unsigned int k0 = 0; k0 < NHP_DA_GROUP; k0 += 16) {
            double r[16];
#pragma unroll
            for (unsigned int k = 0; k < 16; ++k)
                r[k] = __hip_atomic_load(&a.partial[(size_t)(gfirst + (k0 + k < gsize ? k0 + k : 0)) * N + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (unsigned int k = 0; k < 16; ++k) sum += k0 + k < gsize ? r[k] : 0.0;
        }
        atomicAdd(&gout[c], sum);                                   // (the row the next launch reads; zeroed two launches ago)
    }
    if (tid == 0) __hip_atomic_store(&a.tick[32 * grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DADJ_ST(5);
}

// (A cooperative launch holding every workgroup for all N steps -- entries and λ in registers, the decision announced through
// a polled counter, the next step's tables fetched meanwhile -- was built and measured: 31.8 ms per sweep against 24.9.  With
// 12 entries a thread the 128 registers do not hold words + λ + x, so a flip is carried by recomputing x for every entry of
// a wave that holds one flipped column (13.6 µs a step, as much as the entries' pass itself), and the pass spills
// (18.7 µs against 13.5); stamps in profiles/README.md.  Removed.)
extern "C" nhp_status nhp_disc_resample_adjacency(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                                  const double *W, const double *theta, double *A, double dt,
                                                  const double *rho_matrix, double rho, const double *u,
                                                  uint64_t seed, uint64_t step, double *n_links)
{
    if (!ctx || !ds || !A) return NHP_EINVAL;
    if (!rho_matrix && !(rho >= 0.0 && rho <= 1.0)) { nhp_set_error(ctx, "link probability must lie in [0, 1]"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B, TN = (size_t)ds->T * N;
    const size_t nocc = (size_t)(ds->nocc_pad > 0 ? ds->nocc_pad : 4);
    const int nwg = ds->da_nspans;                                     // the spans were cut when the dataset was made (NHP_DADJ_SPANS)
    int th = 512;                                                      // (1024 threads: 64 registers each -- the loop spills, 47 ms against 25)
    if (const char *cs = getenv("NHP_DADJ_THREADS")) th = atoi(cs);
    if (th != 256 && th != 1024) th = 512;
    const int ngrp = (nwg + NHP_DA_GROUP - 1) / NHP_DA_GROUP;
    const size_t gsz = std::max<size_t>(2 * B * (size_t)NHP_DA_SPAN, std::max<size_t>(N, 1024));
    const size_t lds_base = 8 * (gsz + 3 * N), lds_v = 8 * N * B;
    const bool vlds = lds_base + lds_v <= 78 * 1024 && !(getenv("NHP_DADJ_VLDS") && atoi(getenv("NHP_DADJ_VLDS")) == 0);   // (two workgroups per CU)
    const size_t lds = lds_base + (vlds ? lds_v : 0);
    if (lds > 160 * 1024 - 64) {
        nhp_set_error(ctx, "resample_adjacency: N = %d, B = %d exceed the LDS budget", ds->N, ds->B);
        return NHP_ENOTIMPL;
    }
    // scratch after stage_bump's own block: λ (T·N, for the initial gather) | λ_occ | partial | group partials (two halves) |
    // V (all rows) | A's rows | logit u | log ρ | log(1-ρ) | Σ_t x_t | u | ρ | tickets
    const size_t extra = TN + nocc + 2 + (size_t)nwg * N + std::max<size_t>(2 * (size_t)ngrp, 3) * N + NN * B + 5 * NN + 2 * NN + 4 * (size_t)(2 + ngrp) * 4 + 16;
    double *E, *base, *x;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E, &base, extra, &x, 0));
    double *dlam = x; x += TN;
    double *lam_occ = x + (((uintptr_t)x & 15) ? 1 : 0); x += nocc + 2;      // (16-byte aligned: read as double2)
    double *partial = x; x += (size_t)nwg * N;
    double *gpartial = x; x += std::max<size_t>(2 * (size_t)ngrp, 3) * N;               // (three rows of N are used: the steps' sums, p mod 3)
    double *Vall = x; x += NN * B;
    double *AT = x; x += NN;
    double *LU = x; x += NN;
    double *LR1 = x; x += NN;
    double *LR2 = x; x += NN;
    double *SX = x; x += NN;
    double *d_u = x; x += NN;
    double *d_rho = x; x += NN;
    unsigned int *tick = reinterpret_cast<unsigned int *>(x);            // 32 words per group ticket
    // stage_bump left W, θ, A on the device just before the extra block: recover the pointers
    double *dW = base + 2 * N, *dth = dW + NN, *dA = dth + NN * B;
    hipStream_t st = ctx->stream;
    if (u) NHP_HIP(ctx, hipMemcpyAsync(d_u, u, 8 * NN, hipMemcpyHostToDevice, st));
    if (rho_matrix) NHP_HIP(ctx, hipMemcpyAsync(d_rho, rho_matrix, 8 * NN, hipMemcpyHostToDevice, st));
    // λ under the current A at the occupied bins: from the entry lists themselves (k_dadj_lambda0) where a thread's share fits
    // its registers, else the T x N intensity GEMM and a gather (NHP_DADJ_LAMBDA0=0: always the GEMM)
    // (the list route skips absent links: measured against the GEMM at link densities 0.2 / 0.5 / 0.8 / 1.0: -0.8 / -2.6 / +1.6 / +1.4 ms)
    size_t links = 0;
    for (size_t i = 0; i < NN; ++i) links += A[i] != 0.0;
    const int l0_env = getenv("NHP_DADJ_LAMBDA0") ? atoi(getenv("NHP_DADJ_LAMBDA0")) : -1;        // 0 | 1: never | whenever possible
    const bool lam_pass = ds->d_occ_pack && vlds && (ds->B == 8 || ds->B == 4) && ds->da_max_entries <= 12 * 512 && 2 * 512 >= (int)N &&
                          l0_env != 0 && (l0_env == 1 || (double)links <= 0.6 * (double)NN);
    if (!lam_pass) {
        NHP_TRY(nhp_disc_launch_intensity(ctx, ds, E, base, lambda0 == nullptr, dlam));
        hipLaunchKernelGGL(k_dadj_gather, dim3((unsigned)((nocc + 255) / 256)), dim3(256), 0, st, dlam, ds->d_occ_t, ds->d_occ_c,
                           ds->nocc_pad, ds->T, lam_occ);
    }
    hipLaunchKernelGGL(k_dadj_tables, dim3((unsigned)((N + 15) / 16), (unsigned)((N + 15) / 16)), dim3(256), 0, st, ds->N, ds->B, dt, dW, dth, dA,
                       rho_matrix ? d_rho : nullptr, rho, u ? d_u : nullptr, seed, step, Vall, AT, LU, LR1, LR2, ds->d_convsum, SX);
    NHP_HIP(ctx, hipMemsetAsync(tick, 0, 4 * (32 * (size_t)(1 + ngrp) + 4), st));
    NHP_HIP(ctx, hipMemsetAsync(gpartial, 0, 8 * 3 * N, st));
    NHP_HIP(ctx, hipGetLastError());
    nhp_dadj_args a{};
    a.N = ds->N; a.B = ds->B; a.nspans = nwg; a.gsz = (int)gsz; a.T = ds->T; a.conv = ds->d_conv; a.occ_pack = ds->d_occ_pack;
    a.occ_t = ds->d_occ_t; a.occ_c = ds->d_occ_c; a.occ_off = ds->d_occ_off; a.span_t = ds->d_span_t; a.occ_s = ds->d_occ_s;
    a.Vall = Vall; a.AT = AT; a.LU = LU; a.LR1 = LR1; a.LR2 = LR2; a.SX = SX; a.lam_occ = lam_occ;
    a.partial = partial; a.gpartial = gpartial; a.tick = tick; a.A = dA;
    if (lam_pass) {
        const size_t l0 = 8 * ((size_t)B * NHP_DA_SPAN + N + N * B);
        if (ds->B == 8) {
            if (l0 > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_dadj_lambda0<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
            hipLaunchKernelGGL((k_dadj_lambda0<8, 512>), dim3((unsigned)nwg), dim3(512), l0, st, a, (const double *)base, (const double *)(lambda0 ? nullptr : ds->d_baseT));
        } else {
            if (l0 > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_dadj_lambda0<4, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
            hipLaunchKernelGGL((k_dadj_lambda0<4, 512>), dim3((unsigned)nwg), dim3(512), l0, st, a, (const double *)base, (const double *)(lambda0 ? nullptr : ds->d_baseT));
        }
        NHP_HIP(ctx, hipGetLastError());
    }
#ifdef DADJ_STAMP
    unsigned long long *d_st = nullptr;
    const int st_step = getenv("NHP_DADJ_STAMP_STEP") ? atoi(getenv("NHP_DADJ_STAMP_STEP")) : 100;
    if (getenv("NHP_DADJ_STAMPS") && hipMalloc((void **)&d_st, 64 * (size_t)nwg) == hipSuccess) (void)hipMemsetAsync(d_st, 0, 64 * (size_t)nwg, st);
#define DADJ_STAMP_ARG a.stamps = (p == st_step ? d_st : nullptr)
#else
#define DADJ_STAMP_ARG (void)0
#endif
#define DADJ_GO(BT, PK, VL, TH_)                                                                                                    \
    do {                                                                                                                           \
        if (lds > 64 * 1024)                                                                                                       \
            (void)hipFuncSetAttribute((const void *)k_dadj_step<BT, PK, VL, TH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        for (int p = 0; p <= ds->N; ++p) {                         /* launch N: the last row's decision alone */                  \
            DADJ_STAMP_ARG;                                                                                                        \
            hipLaunchKernelGGL((k_dadj_step<BT, PK, VL, TH_>), dim3(p < ds->N ? (unsigned)nwg : 1u), dim3(TH_), lds, st, p, a);    \
        }                                                                                                                          \
    } while (0)
#define DADJ_TH(BT, PK, VL) do { if (th == 256) DADJ_GO(BT, PK, VL, 256); else if (th == 512) DADJ_GO(BT, PK, VL, 512); else DADJ_GO(BT, PK, VL, 1024); } while (0)
#define DADJ_V(BT, PK) do { if (vlds) DADJ_TH(BT, PK, true); else DADJ_TH(BT, PK, false); } while (0)
    const bool pk = ds->d_occ_pack != nullptr;
    if (ds->B == 8) { if (pk) DADJ_V(8, true); else DADJ_V(8, false); }
    else if (ds->B == 4) { if (pk) DADJ_V(4, true); else DADJ_V(4, false); }
    else { if (pk) DADJ_V(0, true); else DADJ_V(0, false); }
#undef DADJ_V
#undef DADJ_TH
#undef DADJ_GO
#ifdef DADJ_STAMP
    if (d_st) {
        std::vector<unsigned long long> h(8 * (size_t)nwg);
        (void)hipMemcpyAsync(h.data(), d_st, 64 * (size_t)nwg, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        if (FILE *f = fopen(getenv("NHP_DADJ_STAMPS"), "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        (void)hipFree(d_st);
    }
#endif
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(A, dA, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    if (n_links) { double s = 0.0; for (size_t i = 0; i < NN; ++i) s += A[i]; *n_links = s; }
    return NHP_OK;
}
