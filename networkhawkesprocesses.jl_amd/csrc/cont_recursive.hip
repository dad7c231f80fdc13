// Recursive exponential log-likelihood (reference: recursive_loglikelihood,
// src/continuous.jl:241-276, network twin :407-442).
//
// What the reference's O(M*N) recursion computes, for every event i on child node c:
//   λ_i = λ0_c(t_i) + Σ_p (A[p,c] W[p,c] θ[p,c]) * S_pc(t_i),
//   S_pc(t) = Σ_{j<i, n_j=p, t_j>0} exp(-θ[p,c] (t - t_j))
// i.e. ALL earlier events are parents -- Δtmax is ignored -- except events at
// exactly t = 0.0, which the `parenttimes > 0.0` seen-flag drops (D9); the integral term of
// the network twin is NOT masked by A (D7).  The reference walks events serially and keeps an
// N x N state; one event touches one row (update) and one column (read) of it.
//
// CDNA4 mapping: the state is partitioned by COLUMN.  Workgroup c owns S_{.,c} (N doubles in
// LDS) plus column c of θ and of A∘W∘θ, and visits only the children of node c (node
// buckets).  Between two consecutive children k-1 and k it folds the events of the segment
// [idx_{k-1}, idx_k) -- contiguous in the time-ordered arrays, so the loads are coalesced --
// into an accumulator with one exp and one LDS atomic each, then decays the state to t_k and
// takes the dot product with the weights (one exp per parent node), lanes across parent
// nodes.  Exp count per evaluation: M*N (segments) + M*N (decays) = the reference's 2*M*N,
// but N workgroups wide and 256 lanes deep instead of serial.
#include <algorithm>

#include "nhp_internal.h"
#include "nhp_math.h"

#define NHP_RING 64   // children whose wave partials are buffered before the log pass
#define NHP_REC_PAIR_SLOT 70   // a free 128-byte line of ctx->d_counter (lines 0..64 are the ll kernel's tickets)

__device__ __forceinline__ double rec_baseline(const nhp_cont_args &a, int c, double t)
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

// Layout of one workgroup (column c): thread `tid` owns parent nodes p = tid + 256 q; their state
// S_pc, θ_pc and (a·w·θ)_pc live in registers.  LDS holds θ[·,c] (gathered by node in the segment
// pass) and TWO segment accumulators, so that folding segment k+1 and consuming segment k share
// one barrier interval: one barrier per child instead of two, and two independent instruction
// streams for the scheduler to interleave.
// BLOCK threads, PQ parent nodes per thread (N <= BLOCK * PQ); the launcher picks 256 x 8 up to
// N = 2048 and 512 x 8 up to 4096.
template <int BLOCK, int PQ>
__global__ __launch_bounds__(BLOCK) void k_recursive(nhp_cont_args a, double *__restrict__ partials)
{
    constexpr int WAVES = BLOCK / 64, REC_PQ = PQ, NP = BLOCK * PQ;   // NP >= N padded state slots: no bounds branches
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);              // [8]
    double *ring = red + 8;                                      // [2][NHP_RING * WAVES]
    double *th = ring + 2 * NHP_RING * WAVES;                    // [NP] θ[p,c]
    double *acc0 = th + NP;                                      // [NP] segment accumulators (double-buffered)
    double *acc1 = acc0 + NP;

    const int c = a.col_begin + blockIdx.x, N = a.N, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    double S[REC_PQ], thr[REC_PQ], wthr[REC_PQ];
    double integ = 0.0;
#pragma unroll
    for (int q = 0; q < REC_PQ; ++q) {
        const int p = tid + q * BLOCK;
        S[q] = 0.0; thr[q] = 0.0; wthr[q] = 0.0;
        th[p] = 0.0; acc0[p] = 0.0; acc1[p] = 0.0;               // slots p >= N stay (θ, aWθ, S) = 0
        if (p < N) {
            const size_t k = (size_t)p + (size_t)c * N;
            const double w = a.W[k];
            const double weff = a.A ? a.A[k] * w : w;
            const double t = a.p1[k];
            thr[q] = t; wthr[q] = weff * t;
            th[p] = t;
            integ += a.cnt[p] * w;                               // unmasked: src/continuous.jl:247,413
        }
    }
    __syncthreads();

    const int kb = a.boff[c], ke = a.boff[c + 1];
    double logsum = 0.0;

    // fold events into acc, referenced to time tk  (t_j > 0: the D9 seen-flag).  The FU exponentials of a thread are
    // evaluated unconditionally (clamped records) and only the atomics are predicated: one basic block, so the
    // independent polynomial chains interleave instead of running one masked block after the other.
    constexpr int FU = 4;                                        // events per thread in flight together
    auto fold_regs = [&](double *acc, const nhp_event *e, int j0, int je, double tk) {
        double v[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) v[u] = nhp_exp_neg_ll(-(th[e[u].node] * (tk - e[u].t)));
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (j0 + u * BLOCK < je && e[u].t > 0.0) atomicAdd(&acc[e[u].node], v[u]);
    };
    auto fold = [&](double *acc, int jb, int je, double tk) {
        for (int j0 = jb + tid; j0 < je; j0 += FU * BLOCK) {
            nhp_event e[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) e[u] = a.ev[j0 + u * BLOCK < je ? j0 + u * BLOCK : j0];
            fold_regs(acc, e, j0, je, tk);
        }
    };

    // child records (time, index) are kept in scalars and fetched two ahead, so their latency never
    // sits on the loop (a struct copy here makes hipcc spill the records to scratch)
    double ch_t = 0.0, nx_t = 0.0;
    int ch_idx = 0, nx_idx = 0;
    if (kb < ke) { ch_t = a.child[kb].t; ch_idx = a.child[kb].idx; nx_t = ch_t; nx_idx = ch_idx; }
    if (kb + 1 < ke) { nx_t = a.child[kb + 1].t; nx_idx = a.child[kb + 1].idx; }
    if (kb < ke) fold(acc0, 0, ch_idx, ch_t);
    __syncthreads();
    double prev_t = ch_t;                                        // first child: gap 0, exp(-0) = 1 on S = 0
    double pending = 0.0;                                        // child k-1's lane partial: reduced under child k's math
    for (int k = kb; k < ke; ++k) {
        const int par = (k - kb) & 1;
        double *accA = par ? acc0 : acc1;                        // filled now, consumed next iteration
        double *accB = par ? acc1 : acc0;                        // filled last iteration, consumed now
        const int kn = k + 2 < ke ? k + 2 : ke - 1;
        const double nn_t = a.child[kn].t;
        const int nn_idx = a.child[kn].idx;
        // the first FU events per thread of the next segment are requested before the decay and folded after it, so
        // their latency hides under the decay's exponentials; longer segments finish in the ordinary loop
        const bool more = k + 1 < ke;
        const int fb = ch_idx + tid, fe = more ? nx_idx : ch_idx;
        nhp_event pe[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) pe[u] = a.ev[fb + u * BLOCK < fe ? fb + u * BLOCK : 0];
        // decay the state to t_k, merge segment k, dot with the weights
        const double gap = ch_t - prev_t;
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < REC_PQ; ++q) {
            const int p = tid + q * BLOCK;
            double s = S[q] * nhp_exp_neg_ll(-(thr[q] * gap));
            s += accB[p];
            accB[p] = 0.0;
            S[q] = s;
            part += wthr[q] * s;
        }
        // child k-1's reduction: independent of everything above, so its cross-lane latency is covered
        if (k > kb) {
            const double r = nhp_wave_sum(pending);
            const int o = k - 1 - kb;
            if (lane == 0) ring[(((o / NHP_RING) & 1) * NHP_RING + (o & (NHP_RING - 1))) * WAVES + wave] = r;
        }
        pending = part;
        fold_regs(accA, pe, fb, fe, nx_t);
        if (more) fold(accA, ch_idx + FU * BLOCK, nx_idx, nx_t);
        __syncthreads();
        // logs of a full half of the ring (children k-64 .. k-1), one lane each
        const int done = k - kb;                                 // children whose partials are in the ring
        if (done > 0 && (done & (NHP_RING - 1)) == 0) {
            const int half = ((done - 1) / NHP_RING) & 1;
            if (tid < NHP_RING) {
                const double tk = a.child[k - NHP_RING + tid].t;
                double lam = rec_baseline(a, c, tk);
                for (int w = 0; w < WAVES; ++w) lam += ring[(half * NHP_RING + tid) * WAVES + w];
                logsum += nhp_log(lam);
            }
        }
        prev_t = ch_t;
        ch_t = nx_t; ch_idx = nx_idx;
        nx_t = nn_t; nx_idx = nn_idx;
    }
    if (kb < ke) {                                               // the last child's reduction and the ring's remainder
        const int o = ke - 1 - kb;
        const double r = nhp_wave_sum(pending);
        if (lane == 0) ring[(((o / NHP_RING) & 1) * NHP_RING + (o & (NHP_RING - 1))) * WAVES + wave] = r;
        __syncthreads();
        const int slot = o & (NHP_RING - 1), half = (o / NHP_RING) & 1;
        if (tid <= slot) {
            const double tk = a.child[ke - 1 - slot + tid].t;
            double lam = rec_baseline(a, c, tk);
            for (int w = 0; w < WAVES; ++w) lam += ring[(half * NHP_RING + tid) * WAVES + w];
            logsum += nhp_log(lam);
        }
    }
    const double blk = nhp_block_sum_n<WAVES>(logsum, red);
    const double blk_int = nhp_block_sum_n<WAVES>(integ, red);
    if (tid == 0) {
        partials[2 * (size_t)blockIdx.x] = blk;
        partials[2 * (size_t)blockIdx.x + 1] = blk_int;
    }
}

// ---- the recursion without a per-child barrier: one WAVE per (column, part of the parent nodes) ---------------------
// k_recursive above folds a segment with the whole workgroup (rounds of FU·BLOCK events whatever the segment's length:
// with ~N events between two children of a column and rounds of 1024, 1.58 rounds are evaluated for every one needed)
// and then meets at a barrier before the decay: its waves wait 58 % of their time (profiles/README.md).  Here the parent
// nodes are cut into H parts of NP = 64·PQ nodes and wave h of workgroup c owns S_{p,c} for the parents of part h ONLY:
// it folds the events that fell on ITS parents (a per-part, time-ordered event list made once per dataset: data only)
// in rounds of 64 and decays ITS states -- no other wave ever touches its accumulators, so the LDS atomics of a segment and
// the reads of the decay are ordered by the wave's own instruction stream and the child loop has no barrier at all.  The
// H partial intensities of a child meet in a ring (one barrier per NHP_RING children, as above).  Exponentials through
// the 2^(j/64) table with -θ·64/ln 2 held per parent (nhp_exp_neg_tab_scaled).

#define NHP_RECW_TS 66          // row stride of the reduction tile (doubles): 64 lanes + 2 of padding
#define NHP_RECW_LDS(PQ, H) (sizeof(double) * (16 + 64 + 2 * NHP_RING * (size_t)(H) + (size_t)(H) * (2 * 64 * (PQ) + 8 * NHP_RECW_TS)))

template <int PQ, int H>
__global__ __launch_bounds__(64 * H) void k_recursive_waves(nhp_cont_args a, nhp_rec_parts rp, double *__restrict__ partials,
                                                            double *__restrict__ ginv /* [M] by bucket position: 1/λ of every child (gradient pass), or null */)
{
    constexpr int NP = 64 * PQ;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);              // [16]
    double *tab = red + 16;                                      // [64] 2^(j/64)
    double *ring = tab + 64;                                     // [2][NHP_RING][H]
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    double *th = ring + 2 * NHP_RING * H + (size_t)h * (2 * NP + 8 * NHP_RECW_TS);   // [NP] -θ[p,c]·64/ln2 of this wave's parents
    double *acc = th + NP;                                       // [NP] segment accumulator (this wave's alone)
    double *tile = acc + NP;                                     // [8][TS] lane partials of 8 children, summed 8 at a time
    const int c = a.col_begin + blockIdx.x, N = a.N;
    const double K64 = 92.33248261689366;                        // 64 / ln 2

    nhp_exp_tab_init(tab);
    double S[PQ], thr[PQ], wthr[PQ];
    double integ = 0.0;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int pl = lane + 64 * q, p = h * NP + pl;
        S[q] = 0.0; thr[q] = 0.0; wthr[q] = 0.0;
        if (p < N) {
            const size_t k = (size_t)p + (size_t)c * N;
            const double w = a.W[k];
            const double weff = a.A ? a.A[k] * w : w;
            const double t = a.p1[k];
            thr[q] = -(t * K64); wthr[q] = weff * t;
            integ += a.cnt[p] * w;                               // unmasked: src/continuous.jl:247,413
        }
        th[pl] = thr[q];
        acc[pl] = 0.0;
    }
    __syncthreads();                                             // the table (the only LDS shared between the waves besides the ring)

    const nhp_event *ev = rp.ev + rp.poff[h];
    const int32_t *rk = rp.rank + (size_t)h * (size_t)a.M;
    const int kb = a.boff[c], ke = a.boff[c + 1];
    double logsum = 0.0;

    // fold records [j, je) (j = first + lane; `e` = record j, already requested) into acc, referenced to time tk: rounds of
    // 64, the next round's record requested before this round's arithmetic (two rounds per trip, so the request lands in
    // the register the next round reads).  The list carries 192 records of padding: a lane past the end reads a harmless
    // record and only skips the atomic.  (θ gathered from the parameter column in memory instead of LDS, to unload the LDS
    // unit: 2.08 -> 2.61 ms -- a 64-lane random gather costs the texture path more than the LDS read it replaces.)
    auto term = [&](const nhp_event &e, bool live, double tk) {
#ifdef NHP_RECW_POLY_FOLD
        const double v = nhp_exp_neg_ll((th[e.node] * (tk - e.t)) * (1.0 / 92.33248261689366));
#else
        const double v = nhp_exp_neg_tab_scaled(th[e.node] * (tk - e.t), tab);
#endif
        if (live) atomicAdd(&acc[e.node], v);
    };
    auto fold = [&](nhp_event e, int j, int je, double tk) {
        int r = j - lane;
        while (r < je) {
            const nhp_event en = ev[j + 64];
            asm volatile("" ::: "memory");                       // keeps the request ahead of the math
            term(e, j < je, tk);
            r += 64;
            if (r >= je) break;
            e = ev[j + 128];
            asm volatile("" ::: "memory");
            term(en, j + 64 < je, tk);
            r += 64; j += 128;
        }
    };

    double ch_t = 0.0, nx_t = 0.0;
    int ch_r = 0, nx_r = 0;
    if (kb < ke) { ch_t = a.child[kb].t; ch_r = rk[kb]; nx_t = ch_t; nx_r = ch_r; }
    if (kb + 1 < ke) { nx_t = a.child[kb + 1].t; nx_r = rk[kb + 1]; }
    if (kb < ke) fold(ev[lane], lane, ch_r, ch_t);
    double prev_t = ch_t;
    for (int k = kb; k < ke; ++k) {
        const int kn = k + 2 < ke ? k + 2 : ke - 1;
        const double nn_t = a.child[kn].t;
        const int nn_r = rk[kn];
        const bool more = k + 1 < ke;
        const int fj = ch_r + lane, fe = more ? nx_r : ch_r;
        const nhp_event pe = ev[fj];                             // first round of the next segment, requested before the decay
        NHP_RECW_SYNC();                                          // this wave's atomics of segment k have landed
        const double gap = ch_t - prev_t;
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int pl = lane + 64 * q;
#ifdef NHP_RECW_POLY_DECAY
            double s = S[q] * nhp_exp_neg_ll((thr[q] * gap) * (1.0 / 92.33248261689366));
#else
            double s = S[q] * nhp_exp_neg_tab_scaled(thr[q] * gap, tab);
#endif
            s += acc[pl];
            acc[pl] = 0.0;
            S[q] = s;
            part += wthr[q] * s;
        }
        // Σ over the lanes, 8 children at a time: every lane parks its partial (one LDS write per child); after the
        // eighth, lane (child, eighth) adds 8 parked values and three DPP steps finish the sum inside its group of 8 lanes
        const int o = k - kb;
        tile[(o & 7) * NHP_RECW_TS + lane] = part;
        if ((o & 7) == 7 || !more) {
            NHP_RECW_SYNC();
            const int chl = lane >> 3, sg = lane & 7;
            const double *row = tile + chl * NHP_RECW_TS + 8 * sg;
            double s = ((row[0] + row[1]) + (row[2] + row[3])) + ((row[4] + row[5]) + (row[6] + row[7]));
            s = nhp_dpp_add(s, 0);
            s = nhp_dpp_add(s, 1);
            s = nhp_dpp_add(s, 2);
            const int oc = (o & ~7) + chl;
            if (sg == 0 && chl <= (o & 7)) ring[(((oc / NHP_RING) & 1) * NHP_RING + (oc & (NHP_RING - 1))) * H + h] = s;
        }
        fold(pe, fj, fe, nx_t);
        const int done = o + 1;                                  // children whose partials are in the ring or the tile
        if ((done & (NHP_RING - 1)) == 0 || !more) {             // a full half of the ring (or the rest): the only barrier of the loop
            __syncthreads();
            const int n = ((done - 1) & (NHP_RING - 1)) + 1, half = ((done - 1) / NHP_RING) & 1;
            if (tid < n) {
                const double tk = a.child[k + 1 - n + tid].t;
                double lam = rec_baseline(a, c, tk);
                for (int w = 0; w < H; ++w) lam += ring[(half * NHP_RING + tid) * H + w];
                logsum += nhp_log(lam);
                if (ginv) ginv[k + 1 - n + tid] = 1.0 / lam;
            }
        }
        prev_t = ch_t;
        ch_t = nx_t; ch_r = nx_r;
        nx_t = nn_t; nx_r = nn_r;
    }
    const double blk = nhp_block_sum_n<H>(logsum, red);
    const double blk_int = nhp_block_sum_n<H>(integ, red);
    if (tid == 0) {
        partials[2 * (size_t)blockIdx.x] = blk;
        partials[2 * (size_t)blockIdx.x + 1] = blk_int;
    }
}

// ---- the same sum through a truncated window ------------------------------------------------------------
// The recursion adds, for child i, EVERY earlier event j with weight (a·w·θ)·e^{-θ(t_i - t_j)}.  Events older than
//     cut = ln( M · max(w·θ) / (min λ0 · 2^-60) ) / min θ
// contribute, all together, less than 2^-60 of λ_i (λ_i >= λ0_c, at most M of them, each below max(wθ)·e^{-minθ·cut}),
// i.e. nothing an fp64 sum of the kept terms can register.  When that window is short compared with the 2·N
// exponentials per event of the recursion, the windowed kernel (H1) evaluates the full-history sum instead: same
// parents in the same time order, the recursion's two quirks kept (events at exactly t = 0 skipped, D9; integral
// term not masked by A, D7).  The bound is recomputed when the parameters change; without a positive λ0 and θ, or
// with a long window, the O(M·N) recursion below runs.
__global__ __launch_bounds__(256) void k_rec_stats(nhp_cont_args a, double *__restrict__ out /* [3 * gridDim.x] */)
{
    __shared__ double r0[NHP_WAVES], r1[NHP_WAVES], r2[NHP_WAVES];
    const size_t NN = (size_t)a.N * a.N, nl = a.baseline_kind == NHP_BASELINE_HOMOGENEOUS ? (size_t)a.N : (size_t)a.N * a.grid_n;
    double tmin = __builtin_inf(), wmax = 0.0, lmin = __builtin_inf();
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < NN; k += (size_t)gridDim.x * 256) {
        const double th = a.p1[k], wt = a.W[k] * th;
        tmin = th < tmin || th != th ? th : tmin;
        wmax = wt > wmax || wt != wt ? wt : wmax;
    }
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < nl; k += (size_t)gridDim.x * 256) {
        const double l = a.lambda0[k];
        lmin = l < lmin || l != l ? l : lmin;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double t2 = __shfl_xor(tmin, off), w2 = __shfl_xor(wmax, off), l2 = __shfl_xor(lmin, off);
        tmin = t2 < tmin || t2 != t2 ? t2 : tmin;
        wmax = w2 > wmax || w2 != w2 ? w2 : wmax;
        lmin = l2 < lmin || l2 != l2 ? l2 : lmin;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { r0[wave] = tmin; r1[wave] = wmax; r2[wave] = lmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NHP_WAVES; ++w) {
            tmin = r0[w] < tmin || r0[w] != r0[w] ? r0[w] : tmin;
            wmax = r1[w] > wmax || r1[w] != r1[w] ? r1[w] : wmax;
            lmin = r2[w] < lmin || r2[w] != r2[w] ? r2[w] : lmin;
        }
        out[3 * blockIdx.x] = tmin; out[3 * blockIdx.x + 1] = wmax; out[3 * blockIdx.x + 2] = lmin;
    }
}

// child_cut[k] = child[k] with window start = first event with t_j > t_i - cut, never before the events at t = 0
__global__ __launch_bounds__(256) void k_rec_windows(const nhp_child *__restrict__ child, const double *__restrict__ times,
                                                     int64_t M, double cut, int n_zero, nhp_child *__restrict__ child_cut,
                                                     unsigned long long *__restrict__ pairs)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (k < M) {
        nhp_child ch = child[k];
        const double lim = ch.t - cut;
        int lo = n_zero, hi = ch.idx;                       // first j in [n_zero, idx] with times[j] > lim
        if (lo > hi) lo = hi;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (times[mid] > lim) hi = mid; else lo = mid + 1; }
        ch.first = lo;
        child_cut[k] = ch;
        mine = (unsigned long long)(ch.idx - lo);
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(pairs, mine);
}

// cut for this model (cached per parameter version); 0 = no usable bound
static nhp_status rec_cut_for(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *cut)
{
    // the bound uses the dataset's slab statistics as well as the parameters: key it on both (a dataset id, not its size --
    // two datasets of equal length have different crowding)
    if (m->rec_version == m->version && m->rec_ds == ds->uid) { *cut = m->rec_cut; return NHP_OK; }
    NHP_TRY(nhp_dataset_slab_stats(ctx, ds));               // the data's crowding statistics, made once
    const int blocks = 64;
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 3 * (size_t)blocks));
    nhp_cont_args a = nhp_make_args(ds, m);
    hipLaunchKernelGGL(k_rec_stats, dim3(blocks), dim3(256), 0, ctx->stream, a, ctx->d_partials);
    NHP_HIP(ctx, hipGetLastError());
    double h[3 * 64];
    NHP_HIP(ctx, hipMemcpyAsync(h, ctx->d_partials, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tmin = h[0], wmax = h[1], lmin = h[2];
    for (int b = 1; b < blocks; ++b) {
        if (h[3 * b] < tmin || h[3 * b] != h[3 * b]) tmin = h[3 * b];
        if (h[3 * b + 1] > wmax || h[3 * b + 1] != h[3 * b + 1]) wmax = h[3 * b + 1];
        if (h[3 * b + 2] < lmin || h[3 * b + 2] != h[3 * b + 2]) lmin = h[3 * b + 2];
    }
    double c = 0.0;
    if (tmin > 0.0 && lmin > 0.0 && wmax >= 0.0 && tmin < __builtin_inf() && lmin < __builtin_inf() && wmax < __builtin_inf()) {
        // events older than `cut` before a child: at most M of them, or -- counted in slabs of length L going back from
        // the cut, at most slab_max(L) events each, every one at least cut + s·L away -- slab_max(L) / (1 - e^{-θmin·L})
        // "events at distance cut"; the smaller count bounds the dropped tail
        double count = (double)(ds->M > 0 ? ds->M : 1);
        for (size_t q = 0; q < ds->h_slab_len.size(); ++q) {
            const double geo = 1.0 - exp(-tmin * ds->h_slab_len[q]);
            if (geo > 0.0) count = std::min(count, (double)ds->h_slab_max[q] / geo);
        }
        const double mass = count * wmax;
        c = mass > 0.0 ? (log(mass / lmin) + 60.0 * 0.6931471805599453) / tmin : 1e-300;
        if (!(c > 0.0)) c = 1e-300;                          // no excitation at all: an empty window is exact
    }
    m->rec_cut = c;
    m->rec_version = m->version;
    m->rec_ds = ds->uid;
    *cut = c;
    return NHP_OK;
}

static nhp_status nhp_launch_recursive_full(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out);

// *child_cut = the children with truncated-window starts (and *group the lanes-per-child width for them) when the
// full-history sum may be evaluated through a window for this (data, parameters); nullptr when the recursion must run
nhp_status nhp_recursive_window(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, const nhp_child **child_cut,
                                int *group, double recursion_cost)
{
    *child_cut = nullptr;
    static const int mode = getenv("NHP_REC_WINDOW") ? atoi(getenv("NHP_REC_WINDOW")) : 1;     // 0: always the recursion
    if (!mode || ds->M <= 0 || m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_OK;
    double cut = 0.0;
    NHP_TRY(rec_cut_for(ctx, ds, m, &cut));
    // Which is faster?  Measured on MI355X: the recursion (k_recursive_waves) spends ≈0.6 µs + 1.25 ns·N per child of its
    // column (all N <= 1024 columns run at once, so a launch takes that times the largest column: 1.98 ms at N = 1024,
    // M = 1e6), the windowed kernel evaluates ≈9.5·10¹¹ pair terms/s at long windows (≈5.5·10¹¹ below 128 parents) plus
    // ≈30 µs per launch.
    const double rate = ds->t_last > 0.0 ? (double)ds->M / ds->t_last : 0.0;
    const double kc = cut * rate;                          // expected parents per window
    if (!(cut > 0.0) || kc > 8192.0) return NHP_OK;
    int32_t biggest = 1;
    for (int32_t c = 0; c < ds->N; ++c) biggest = std::max(biggest, ds->h_boff[c + 1] - ds->h_boff[c]);
    const double waves_of_columns = (double)((ds->N + 1023) / 1024);
    const double t_recursion = recursion_cost * waves_of_columns * (double)biggest * (0.6e-6 + 1.25e-9 * (double)ds->N);
    auto t_window = [&](double k) { return 30e-6 + (double)ds->M * k / (k >= 128.0 ? 9.5e11 : 5.5e11); };
    if (t_window(kc) >= t_recursion) return NHP_OK;
    nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
    if (!mds->d_child_cut && hipMalloc((void **)&mds->d_child_cut, sizeof(nhp_child) * (size_t)ds->M) != hipSuccess) {
        nhp_set_error(ctx, "out of device memory (recursive windows)");
        return NHP_ENOMEM;
    }
    if (ds->cut_cached != cut) {
        unsigned long long *d_pairs = reinterpret_cast<unsigned long long *>(ctx->d_counter + 32 * (NHP_REC_PAIR_SLOT));
        NHP_HIP(ctx, hipMemsetAsync(d_pairs, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_rec_windows, dim3((unsigned)((ds->M + 255) / 256)), dim3(256), 0, ctx->stream, ds->d_child, ds->d_times,
                           ds->M, cut, (int)ds->n_zero_time, mds->d_child_cut, d_pairs);
        NHP_HIP(ctx, hipGetLastError());
        unsigned long long hp = 0;
        NHP_HIP(ctx, hipMemcpyAsync(&hp, d_pairs, sizeof(hp), hipMemcpyDeviceToHost, ctx->stream));
        NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NHP_HIP(ctx, hipMemsetAsync(d_pairs, 0, sizeof(unsigned long long), ctx->stream));    // counters rest at 0
        ds->cut_cached = cut;
        ds->cut_pairs = (int64_t)hp;
    }
    const double kbar = (double)ds->cut_pairs / (double)ds->M;       // the actual mean window
    if (t_window(kbar) >= t_recursion) return NHP_OK;
    *child_cut = ds->d_child_cut;
    *group = nhp_pick_group(kbar);
    return NHP_OK;
}

nhp_status nhp_launch_recursive(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out)
{
    return nhp_launch_recursive_flags(ctx, ds, m, NHP_LL_RECURSIVE, d_out);
}

nhp_status nhp_launch_recursive_flags(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags, double *d_out)
{
    if (m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const nhp_child *child_cut = nullptr;
    int group = 0;
    if (!(flags & NHP_LL_FULL_RECURSION)) NHP_TRY(nhp_recursive_window(ctx, ds, m, &child_cut, &group));
    if (child_cut) return nhp_launch_windowed_as(ctx, ds, m, child_cut, group, 0, d_out);
    return nhp_launch_recursive_full(ctx, ds, m, d_out);
}

// the per-part event lists of k_recursive_waves for parts of `np` nodes (data only; cached in the dataset)
static nhp_status rec_parts_build(nhp_ctx *ctx, const nhp_cont_dataset *ds, int np, int H)
{
    if (ds->d_rec_ev && ds->rec_np == np && ds->rec_h == H) return NHP_OK;
    nhp_cont_dataset *mds = const_cast<nhp_cont_dataset *>(ds);
    const size_t M = (size_t)ds->M;
    std::vector<double> t(M ? M : 1);
    std::vector<int32_t> node(M ? M : 1);
    NHP_HIP(ctx, hipMemcpyAsync(t.data(), ds->d_times, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipMemcpyAsync(node.data(), ds->d_nodes, sizeof(int32_t) * M, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> poff((size_t)H + 1, 0), seen((size_t)H, 0), rank((size_t)H * (M ? M : 1));
    for (size_t i = 0; i < M; ++i)
        if (t[i] > 0.0) poff[(size_t)(node[i] / np) + 1]++;
    for (int h = 0; h < H; ++h) poff[(size_t)h + 1] += poff[(size_t)h];
    std::vector<nhp_event> ev((size_t)poff[(size_t)H] + 192);
    for (nhp_event &e : ev) { e.t = 0.0; e.node = 0; e.pad = 0; }
    std::vector<int32_t> cur(ds->h_boff.begin(), ds->h_boff.end() - 1);       // bucket position of the next child of each node
    for (size_t i = 0; i < M; ++i) {
        const size_t k = (size_t)cur[(size_t)node[i]]++;
        for (int h = 0; h < H; ++h) rank[(size_t)h * M + k] = seen[(size_t)h];
        if (t[i] > 0.0) {
            const int h = node[i] / np;
            nhp_event &e = ev[(size_t)poff[(size_t)h] + (size_t)seen[(size_t)h]++];
            e.t = t[i]; e.node = node[i] - h * np;
        }
    }
    (void)hipFree(mds->d_rec_ev); (void)hipFree(mds->d_rec_poff); (void)hipFree(mds->d_rec_rank);
    mds->d_rec_ev = nullptr; mds->d_rec_poff = nullptr; mds->d_rec_rank = nullptr; mds->rec_np = 0; mds->rec_h = 0;
    if (hipMalloc((void **)&mds->d_rec_ev, sizeof(nhp_event) * ev.size()) != hipSuccess ||
        hipMalloc((void **)&mds->d_rec_poff, sizeof(int32_t) * poff.size()) != hipSuccess ||
        hipMalloc((void **)&mds->d_rec_rank, sizeof(int32_t) * rank.size()) != hipSuccess) {
        nhp_set_error(ctx, "out of device memory (recursion: per-part event lists)");
        return NHP_ENOMEM;
    }
    NHP_HIP(ctx, hipMemcpyAsync(mds->d_rec_ev, ev.data(), sizeof(nhp_event) * ev.size(), hipMemcpyHostToDevice, ctx->stream));
    NHP_HIP(ctx, hipMemcpyAsync(mds->d_rec_poff, poff.data(), sizeof(int32_t) * poff.size(), hipMemcpyHostToDevice, ctx->stream));
    NHP_HIP(ctx, hipMemcpyAsync(mds->d_rec_rank, rank.data(), sizeof(int32_t) * rank.size(), hipMemcpyHostToDevice, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the host vectors go out of scope
    mds->rec_np = np; mds->rec_h = H;
    return NHP_OK;
}

// shape (parents per lane, parts) of the wave-partitioned recursion for this dataset, and its lists; *PQ = 0 when the
// shape is not available (NHP_REC_WAVES=0, or N > 4096)
nhp_status nhp_rec_parts_for(nhp_ctx *ctx, const nhp_cont_dataset *ds, int *PQ_out, int *H_out, nhp_rec_parts *rp)
{
    *PQ_out = 0; *H_out = 0;
    static const int waves = getenv("NHP_REC_WAVES") ? atoi(getenv("NHP_REC_WAVES")) : 1;
    if (!waves || ds->N > 4096) return NHP_OK;
    int PQ = ds->N <= 256 ? 1 : (ds->N <= 512 ? 2 : 4);
    int H = 1;
    if (const char *env = getenv("NHP_REC_PARTS")) { int q = 0, hh = 0; if (sscanf(env, "%d,%d", &q, &hh) == 2 && 64 * q * hh >= ds->N) { PQ = q; H = hh; } }
    while (64 * PQ * H < ds->N) H *= 2;
    NHP_TRY(rec_parts_build(ctx, ds, 64 * PQ, H));
    *rp = nhp_rec_parts{ds->d_rec_ev, ds->d_rec_poff, ds->d_rec_rank};
    *PQ_out = PQ; *H_out = H;
    return NHP_OK;
}

// the O(M·N) recursion through k_recursive_waves: log-likelihood -> *d_out and, for the gradient pass, 1/λ of every child
// -> d_ginv (bucket order; may be null).  *launched = false when no shape applies (the caller falls back).
nhp_status nhp_launch_recursive_waves(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out, double *d_ginv,
                                      bool *launched)
{
    *launched = false;
    int PQ = 0, H = 0;
    nhp_rec_parts rp{};
    NHP_TRY(nhp_rec_parts_for(ctx, ds, &PQ, &H, &rp));
    if (!PQ) return NHP_OK;
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->N));
    nhp_cont_args a = nhp_make_args(ds, m);
    const size_t lds = NHP_RECW_LDS(PQ, H);
    *launched = true;
#define NHP_RECW(Q, HH)                                                                                                          \
        do {                                                                                                                     \
            if (lds > 64 * 1024)                                                                                                 \
                (void)hipFuncSetAttribute((const void *)k_recursive_waves<Q, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_recursive_waves<Q, HH>), dim3((unsigned)(ds->col_end - ds->col_begin)), dim3(64 * HH), lds, ctx->stream, \
                               a, rp, ctx->d_partials, d_ginv);                                                                  \
        } while (0)
    NHP_REC_SHAPES(NHP_RECW, *launched = false);
#undef NHP_RECW
    if (!*launched) return NHP_OK;
    NHP_HIP(ctx, hipGetLastError());
    return nhp_launch_finalize(ctx, a, ds->col_end - ds->col_begin, d_out);
}

static nhp_status nhp_launch_recursive_full(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out)
{
    if (m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    if (ds->N > 4096) { nhp_set_error(ctx, "recursive ll: n_nodes = %d > 4096 not supported", ds->N); return NHP_ENOTIMPL; }
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->N));
    nhp_cont_args a = nhp_make_args(ds, m);
    // one wave per (column, part of the parents), no barrier in the child loop (k_recursive_waves); NHP_REC_WAVES=0: the
    // workgroup-wide fold + barrier per child (k_recursive).  NHP_REC_PARTS = "PQ,H" forces a shape (tools/kbench.py).
    {
        bool launched = false;
        NHP_TRY(nhp_launch_recursive_waves(ctx, ds, m, d_out, nullptr, &launched));
        if (launched) return NHP_OK;
    }
#define NHP_REC_LAUNCH(B, Q)                                                                                   \
    do {                                                                                                       \
        const size_t lds = sizeof(double) * (8 + 2 * NHP_RING * ((B) / 64) + 3 * (size_t)(B) * (Q));          \
        if (lds > 64 * 1024)                                                                                   \
            (void)hipFuncSetAttribute((const void *)k_recursive<B, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_recursive<B, Q>), dim3((unsigned)(ds->col_end - ds->col_begin)), dim3(B), lds, ctx->stream, a, ctx->d_partials); \
    } while (0)
    // measured at N = 1024, M = 1e6 (first version: 256 x 8 3.16 ms, 512 x 4 3.35, 128 x 8 4.19, 64 x 16 7.35); with the
    // segment loads issued ahead of the decay and the reduction deferred by one child: 256 x 4 2.97 ms
    if (ds->N <= 256) NHP_REC_LAUNCH(256, 1);
    else if (ds->N <= 512) NHP_REC_LAUNCH(256, 2);
    else if (ds->N <= 1024) NHP_REC_LAUNCH(256, 4);
    else if (ds->N <= 2048) NHP_REC_LAUNCH(256, 8);
    else NHP_REC_LAUNCH(512, 8);
#undef NHP_REC_LAUNCH
    NHP_HIP(ctx, hipGetLastError());
    return nhp_launch_finalize(ctx, a, ds->col_end - ds->col_begin, d_out);
}
