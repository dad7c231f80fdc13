// Discrete-time path: basis convolution, intensity / Poisson log-likelihood and the
// mean-field VB step (reference: convolve src/discrete.jl:146-151; basis
// src/impulses.jl:321-335; intensity src/discrete.jl:115-129 with bump :381-385,:511-516;
// loglikelihood :91-102; update! :369-375 = update_parents src/parents.jl:136-177 + the three
// component updates src/baselines.jl:444-456, src/weights.jl:70-97, src/impulses.jl:355-375).
//
// The reference's 4-deep scalar loop for λ and its T x N x (1+NB) responsibility array are one
// dense contraction:  with k = (p,b),  G = Ŝ viewed as T x (N·B)  and  E[k,c] the per-link
// factor,   Z = base ⊕ G·E   (GEMM-1, T x NB x N).  VB needs a second one,
// Γ = E ⊙ (Gᵀ·R) with R = data/Z (GEMM-2, NB x T x N); `u` (1.68 TB at N=512, B=8, T=1e5) is
// never formed.  This IS a dense GEMM, so it runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): 128x128 block tile, 4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles
// (64 fp64 accumulators per lane), BK = 16 staged through padded LDS images that make every
// fragment read conflict-free, next tile prefetched into registers during the MFMAs.
#include <math.h>

#include <algorithm>

#include "nhp_internal.h"
#include "nhp_math.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#ifndef BK
#define BK 16
#endif
#define EPT (BK / 2)    // staged elements per thread and operand: 128 x BK tile / 256 threads
#define A_MC_LD (BM + 16)     // m-contiguous image: row stride ≡ 128 B (mod 256) -> kk rows hit disjoint banks
#define KC_LD (BK + 2)        // k-contiguous image: row stride = 2 (mod 32) bank pairs -> 16 lanes x 2 kk conflict-free

enum { EPI_INTENSITY = 0, EPI_LOGLIK = 1, EPI_VB_Z = 2, EPI_SLAB = 3, EPI_GRAD = 4 };   // EPI_GRAD = EPI_LOGLIK + EPI_VB_Z in one pass

struct gemm_args {
    const double *A; size_t lda;
    const double *B; size_t ldb;
    int M, N;                 // output shape
    int K;                    // reduction length
    int k_chunk;              // reduction elements per grid.z slice
    // epilogue operands
    const double *base;       // [N]   additive per-column term (λ0·dt or e0)
    const double *baseT;      // [M x N] additive per-element term (time-varying baseline); overrides base if non-null
    const double *dataT;      // [M x N] counts as f64, t fastest
    double *out;              // EPI_INTENSITY: λ [M x N]; EPI_VB_Z: R [M x N]; EPI_SLAB: slabs [z][M x N]
    double *partials;         // EPI_LOGLIK: [2 * blocks]; EPI_VB_Z / EPI_GRAD: column partials [rowBlocks][N]
    double *partials2;        // EPI_GRAD: the log-likelihood partials [2 * blocks]
};

// C[m,n] = Σ_k Aop[m,k]·B[k + n·ldb];  A_MCONTIG: Aop[m,k] = A[m + k·lda], else A[k + m·lda].
// TBM = rows of the output tile: 128, or 160 for the m-contiguous GEMM-1 when that fills the last round of workgroups
// better (gemm1_tile_m): 5 instead of 4 MFMA row-fragments per wave, everything else alike.
// WHOLE: every tile of every workgroup is inside the matrices and the reduction chunk is whole BK-tiles (the host checks):
// the main loop is then branch-free and scheduled instruction by instruction (tile_whole).
template <bool A_MCONTIG, int EPI, int TBM = BM, bool WHOLE = false>
__global__ __launch_bounds__(256, 2) void k_gemm_f64(gemm_args g)      // 2 waves/SIMD: <= 256 VGPR+AGPR
{
    static_assert(TBM % 32 == 0 && (A_MCONTIG || TBM == BM), "the k-contiguous A staging covers 128 rows");
    static_assert(TBM == BM || BK == 16, "wide tiles are staged one 16-row k-slab at a time");
    constexpr int MI = TBM / 32;                                   // MFMA row fragments per wave (2 x 2 waves)
    constexpr int MS = TBM / 16;                                   // 16-row m-slots of the staged A tile
    constexpr int EA = A_MCONTIG ? MS * (BK / 16) : EPT;            // staged A elements per thread
    constexpr int A_LD = TBM + 16;                                 // m-contiguous image: row stride ≡ 128 B (mod 256)
    extern __shared__ __align__(16) double gsm[];
    // two stages of {A image, B image}: tile k+1 is written while tile k is read, ONE barrier per tile.  The epilogue's small
    // reduction arrays reuse stage 0 after the last tile's barrier.
    constexpr int SA = A_MCONTIG ? BK * A_LD : BM * KC_LD, SB = BN * KC_LD, STAGE = SA + SB;
    double *red = gsm;                                             // [NHP_WAVES]
    double(*wcol)[64] = reinterpret_cast<double(*)[64]>(red + NHP_WAVES);   // [NHP_WAVES][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kk = lane >> 4;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Remap the linear id
    // so that every XCD owns a CONTIGUOUS run of logical tiles: the n-tiles that share an A panel
    // then hit the same L2 instead of fetching the panel once per XCD (bijective for any grid).
    const unsigned nbk = gridDim.x * gridDim.y, lin = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned xq = nbk / 8, xr = nbk % 8, xcd = lin % 8, pos = lin / 8;
    const unsigned logical = xcd * xq + (xcd < xr ? xcd : xr) + pos;
    const int bx = (int)(logical % gridDim.x), by = (int)(logical / gridDim.x);
    const int m0 = by * TBM, n0 = bx * BN;                    // n-tiles fastest: tiles sharing an A panel are adjacent
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);

    v4d acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // per-thread staging coordinates: 8 consecutive elements along the contiguous dimension
    // m-contig A: thread (k = tid/16 (+16 for the second half of a BK=32 tile), m = tid%16 + 16 e): at fixed e a wave covers 4 k-rows x 128 B,
    // coalesced in HBM and conflict-free as ds_write_b64 (consecutive lanes -> consecutive doubles)
    const int a_k = A_MCONTIG ? tid >> 4 : (tid & 1) * EPT;        // m-contig: k row;  k-contig: k offset
    const int a_m = A_MCONTIG ? (tid & 15) : tid >> 1;             // m-contig: m offset; k-contig: m row
    const int b_k = (tid & 1) * EPT, b_n = tid >> 1;
    double ra[EA], rb[EPT];

    // Per-thread operand pointers advance by one tile per iteration; the 8 elements of a thread sit
    // at compile-time offsets from them, so the loop carries no 64-bit index arithmetic (with one
    // wave per SIMD every VALU instruction is time taken from the MFMA stream).
    const double *pa = A_MCONTIG ? g.A + ((size_t)(m0 + a_m) + (size_t)(kbeg + a_k) * g.lda)
                                 : g.A + ((size_t)(kbeg + a_k) + (size_t)(m0 + a_m) * g.lda);
    const double *pb = g.B + ((size_t)(kbeg + b_k) + (size_t)(n0 + b_n) * g.ldb);
    const size_t a_step = A_MCONTIG ? (size_t)BK * g.lda : (size_t)BK;
    unsigned a_ok = 0;                       // bit e: row of element e is inside the matrix
#pragma unroll
    for (int e = 0; e < (A_MCONTIG ? MS : 8); ++e) a_ok |= ((A_MCONTIG ? m0 + a_m + 16 * e : m0 + a_m) < g.M ? 1u : 0u) << e;   // bit e: m-slot e
    const bool b_ok = n0 + b_n < g.N;

    // Interior tiles (every row, column and k of the tile inside the matrix -- all of them at the benchmark shape) load
    // unconditionally: a predicated load compiles to an exec-mask branch of its own, ~20 of them per tile on the path of a
    // wave whose every non-MFMA cycle is taken from the matrix pipe.
    const bool interior = m0 + TBM <= g.M && n0 + BN <= g.N;
    auto load_tiles = [&](int k0) {
        const bool full = k0 + BK <= kend;   // only the last tile of a ragged K needs per-element checks
        if (interior && full) {
#pragma unroll
            for (int e = 0; e < EA; ++e) ra[e] = A_MCONTIG ? pa[16 * (e % MS) + (size_t)(16 * (e / MS)) * g.lda] : pa[e];
#pragma unroll
            for (int e = 0; e < EPT; ++e) rb[e] = pb[e];
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                // m-contig: element e = (m-slot e % MS, k-row a_k + 16 (e / MS)); k-contig: k offset a_k + e
                const bool ka = full || (A_MCONTIG ? k0 + a_k + 16 * (e / MS) : k0 + a_k + e) < kend;
                const bool ma = A_MCONTIG ? ((a_ok >> (e % MS)) & 1u) : (a_ok & 1u);
                ra[e] = ma && ka ? (A_MCONTIG ? pa[16 * (e % MS) + (size_t)(16 * (e / MS)) * g.lda] : pa[e]) : 0.0;
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const bool kb = full || k0 + b_k + e < kend;
                rb[e] = b_ok && kb ? pb[e] : 0.0;
            }
        }
        pa += a_step;
        pb += BK;
    };
    auto store_tiles = [&](double *As, double *Bs) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            if (A_MCONTIG) As[(a_k + 16 * (e / MS)) * A_LD + a_m + 16 * (e % MS)] = ra[e];
            else As[a_m * KC_LD + a_k + e] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) Bs[b_n * KC_LD + b_k + e] = rb[e];
    };
    struct frag { double a[MI], b[4]; };
    auto read_slab = [&](const double *As, const double *Bs, const int ks, frag &f) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm * (TBM / 2) + i * 16 + r16;
            f.a[i] = A_MCONTIG ? As[(ks * 4 + kk) * A_LD + m] : As[m * KC_LD + ks * 4 + kk];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = wn * 64 + i * 16 + r16;
            f.b[i] = Bs[n * KC_LD + ks * 4 + kk];
        }
    };
    auto mfma_slab = [&](const frag &f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };
    // One tile: the MFMAs of stage `cur`; under them, tile k+1 goes from registers to the other stage and tile k+2 is
    // requested from memory, and every slab's fragments are read while the slab before it multiplies.  sched_barrier pins
    // that order: the non-MFMA work sits between MFMA groups of the same wave (the matrix pipe takes 64 cycles per
    // instruction and runs on while the wave issues other work) instead of in a block of its own between two barriers,
    // where only the other workgroup's wave could cover it.
    static_assert(BK == 16, "the tile schedule below is written for four k-slabs");
    auto tile = [&](const int k0, const int cur) {
        const double *Ac = gsm + cur * STAGE, *Bc = Ac + SA;
        double *An = gsm + (cur ^ 1) * STAGE, *Bn = An + SA;
        frag f0, f1;
#if defined(GEMM_PRIO) && GEMM_PRIO == 1
        __builtin_amdgcn_s_setprio(2);
#endif
        read_slab(Ac, Bc, 0, f0);
        read_slab(Ac, Bc, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_slab(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + BK < kend) store_tiles(An, Bn);
        __builtin_amdgcn_sched_barrier(0);
        read_slab(Ac, Bc, 2, f0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_slab(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 2 * BK < kend) load_tiles(k0 + 2 * BK);
        __builtin_amdgcn_sched_barrier(0);
        read_slab(Ac, Bc, 3, f1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_slab(f0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_slab(f1);
#if defined(GEMM_PRIO) && GEMM_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();                       // stage cur^1 written by all, stage cur read by all
    };

#if defined(GEMM_PRIO) && GEMM_PRIO == 2
    if ((lin >> 3) & 1) __builtin_amdgcn_s_setprio(2);
#elif defined(GEMM_PRIO) && GEMM_PRIO == 3
    if ((lin >> 9) & 1) __builtin_amdgcn_s_setprio(2);
#endif
    // The same tile for a workgroup whose every tile is whole (all of them at the benchmark shapes): no branch inside, so
    // the tile is ONE scheduling region and the order below is imposed instruction by instruction -- an LDS write, a global
    // load or a fragment read after each MFMA.  One wave then keeps the matrix pipe busy through its own LDS/global traffic
    // (each of those issues in a few cycles under a 64-cycle MFMA); clustered, they cost the pipe ~3000 idle cycles per
    // 5120-cycle tile whenever the other workgroup's wave on the SIMD was not there to cover them.
    frag fw0;                                                      // slab 0 of the tile about to run (tile_whole)
    auto tile_whole = [&](const int k0, const int cur) {
        const double *Ac = gsm + cur * STAGE, *Bc = Ac + SA;
        double *An = gsm + (cur ^ 1) * STAGE, *Bn = An + SA;
        constexpr int NM = MI * 4;                                 // MFMAs of a k-slab
        constexpr int NW = (EA + 1) / 2 + EPT / 2;                 // LDS writes of a tile, two elements each
        constexpr int NL = EA + EPT;                               // global loads of a tile, one element each
        constexpr int NR = MI + 4;                                 // fragment reads of a k-slab
        static_assert(NW + NL <= 2 * NM && NR <= NM, "the schedule below places one memory operation after each MFMA");
        const bool more = k0 + 2 * BK < kend;                      // tile k0 + 2 BK: past the end, the last tile is loaded again
        const long back_a = more ? 0 : -(long)a_step, back_b = more ? 0 : -(long)BK;
        auto wr = [&](const int n) {                               // n-th LDS write: tile k0 + BK from the staging registers
            if (n < (EA + 1) / 2) {
#pragma unroll
                for (int e = 2 * n; e < 2 * n + 2 && e < EA; ++e) {
                    if (A_MCONTIG) An[(a_k + 16 * (e / MS)) * A_LD + a_m + 16 * (e % MS)] = ra[e];
                    else An[a_m * KC_LD + a_k + e] = ra[e];
                }
            } else {
                const int e = 2 * (n - (EA + 1) / 2);
                Bn[b_n * KC_LD + b_k + e] = rb[e];
                Bn[b_n * KC_LD + b_k + e + 1] = rb[e + 1];
            }
        };
        auto ld = [&](const int n) {                               // n-th global load: tile k0 + 2 BK into the staging registers
            if (n < EA) ra[n] = A_MCONTIG ? pa[back_a + 16 * (n % MS) + (long)(16 * (n / MS)) * (long)g.lda] : pa[back_a + n];
            else rb[n - EA] = pb[back_b + (n - EA)];
        };
        auto rd = [&](const int ks, const int n, frag &f) {        // n-th fragment read of k-slab ks
            if (n < MI) {
                const int m = wm * (TBM / 2) + n * 16 + r16;
                f.a[n] = A_MCONTIG ? Ac[(ks * 4 + kk) * A_LD + m] : Ac[m * KC_LD + ks * 4 + kk];
            } else {
                const int nn = wn * 64 + (n - MI) * 16 + r16;
                f.b[n - MI] = Bc[nn * KC_LD + ks * 4 + kk];
            }
        };
        auto rdn = [&](const int n, frag &f) {                     // n-th fragment read of k-slab 0 of the NEXT tile (the other stage)
            if (n < MI) {
                const int m = wm * (TBM / 2) + n * 16 + r16;
                f.a[n] = A_MCONTIG ? An[kk * A_LD + m] : An[m * KC_LD + kk];
            } else {
                const int nn = wn * 64 + (n - MI) * 16 + r16;
                f.b[n - MI] = Bn[nn * KC_LD + kk];
            }
        };
        // fw0 holds this tile's slab 0 (read under the previous tile's last slab); the barrier sits between slabs 2 and 3 --
        // by then every fragment of this stage has been read and the other stage was written in slab 0 -- so that the next
        // tile's first fragments travel under slab 3 and no tile starts by waiting for LDS
        frag f1;
        constexpr int L0 = NM - NW;                                // global loads placed in slab 0
#pragma unroll
        for (int n = 0; n < NM; ++n) {                             // slab 0: slab 1's fragments and the writes, then loads
            acc[n / 4][n % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw0.a[n / 4], fw0.b[n % 4], acc[n / 4][n % 4], 0, 0, 0);
            if (n < NR) rd(1, n, f1);
            if (n < NW) wr(n);
            else if (n - NW < NL) ld(n - NW);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < NM; ++n) {                             // slab 1: the rest of the loads, then slab 2's fragments
            acc[n / 4][n % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1.a[n / 4], f1.b[n % 4], acc[n / 4][n % 4], 0, 0, 0);
            if (L0 + n < NL) ld(L0 + n);
            else if (L0 + n - NL < NR) rd(2, L0 + n - NL, fw0);
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = more ? pa + a_step : pa;
        pb = more ? pb + BK : pb;
#pragma unroll
        for (int n = 0; n < NM; ++n) {                             // slab 2: slab 3's fragments
            acc[n / 4][n % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(fw0.a[n / 4], fw0.b[n % 4], acc[n / 4][n % 4], 0, 0, 0);
            if (n < NR) rd(3, n, f1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                       // stage cur^1 written by all (slab 0), stage cur read by all (its last reads: slab 2)
#pragma unroll
        for (int n = 0; n < NM; ++n) {                             // slab 3: the next tile's slab 0, from the other stage
            acc[n / 4][n % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1.a[n / 4], f1.b[n % 4], acc[n / 4][n % 4], 0, 0, 0);
            if (n < NR) rdn(n, fw0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    load_tiles(kbeg);
    store_tiles(gsm, gsm + SA);
    __syncthreads();
    if (kbeg + BK < kend) load_tiles(kbeg + BK);
    if (WHOLE) {
        read_slab(gsm, gsm + SA, 0, fw0);
        for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
            tile_whole(k0, 0);
            if (k0 + BK < kend) tile_whole(k0 + BK, 1);
        }
        __syncthreads();                       // (the epilogue's arrays reuse stage 0: the last tile's slab-3 reads are behind us)
    } else {
        for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
            tile(k0, 0);
            if (k0 + BK < kend) tile(k0 + BK, 1);
        }
    }

    // ---- epilogue.  acc[i][j][r] is C[row, col], row = wm*(TBM/2) + i*16 + kk + 4r, col = wn*64 + j*16 + r16
    double t_sum = 0.0, t_sum2 = 0.0;
    double colp[4] = {0.0, 0.0, 0.0, 0.0};
    // Σ s·log λ needs a logarithm only where the bin holds events -- one bin in twenty at the benchmark's rate, yet a wave
    // pays for all 64 lanes whenever one of them does.  Each wave therefore queues its (λ, s) pairs in LDS (ballot +
    // prefix count: the order depends on the data only) and takes the logarithm 64 at a time: ~5 evaluations per wave and
    // tile instead of 80 (3 % of the kernel).  The queue lives in the stages, free after the last tile's barrier.
    double *qlam = gsm + 512 + wave * 256, *qs = qlam + 128;       // [128] each, per wave
    int qn = 0;                                                     // wave-uniform fill
    auto log_queue_push = [&](const bool nz, const double lam, const double sv) {
        const unsigned long long m = __ballot(nz);
        if (m == 0ull) return;
        const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (nz) { qlam[pos] = lam; qs[pos] = sv; }
        qn += __popcll(m);
        if (qn >= 64) {
            NHP_LDS_SYNC();
            t_sum += qs[lane] * nhp_log(qlam[lane]);
            const bool mv = lane < qn - 64;
            const double ml = mv ? qlam[64 + lane] : 0.0, ms = mv ? qs[64 + lane] : 0.0;
            NHP_LDS_SYNC();
            if (mv) { qlam[lane] = ml; qs[lane] = ms; }
            qn -= 64;
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r16;
        const double base_c = (EPI != EPI_SLAB && col < g.N && !g.baseT) ? g.base[col] : 0.0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (TBM / 2) + i * 16 + kk + 4 * r;
                double sv = 0.0, lamv = 1.0;                       // (EPI_LOGLIK / EPI_GRAD: this element's count and intensity)
                if (row < g.M && col < g.N) {
                    const size_t o = (size_t)row + (size_t)col * g.M;
                    const double v = acc[i][j][r];
                    const double base = (EPI != EPI_SLAB && g.baseT) ? g.baseT[o] : base_c;
                    if (EPI == EPI_INTENSITY) {
                        g.out[o] = base + v;
                    } else if (EPI == EPI_LOGLIK) {
                        // log pdf(Poisson(λ), s) = xlogy(s, λ) - λ - loggamma(s+1); the data-only
                        // Σ loggamma(s+1) is hoisted to the host
                        const double lam = base + v;
                        sv = g.dataT[o]; lamv = lam;
                        t_sum2 += lam;
                    } else if (EPI == EPI_VB_Z) {
                        const double rr = g.dataT[o] / (base + v);
                        g.out[o] = rr;
                        colp[j] += rr;
                    } else if (EPI == EPI_GRAD) {
                        const double lam = base + v, s = g.dataT[o], rr = s / lam;
                        g.out[o] = rr;
                        colp[j] += rr;
                        sv = s; lamv = lam;
                        t_sum2 += lam;
                    } else {
                        g.out[(size_t)blockIdx.z * (size_t)g.M * g.N + o] = v;
                    }
                }
                if (EPI == EPI_LOGLIK || EPI == EPI_GRAD) log_queue_push(sv != 0.0, lamv, sv);
            }
    }
    if (EPI == EPI_LOGLIK || EPI == EPI_GRAD) {
        NHP_LDS_SYNC();
        if (lane < qn) t_sum += qs[lane] * nhp_log(qlam[lane]);    // what is left in the wave's queue
        __syncthreads();                                           // (the queues overlap nothing of red / wcol, but keep the phases apart)
        const double s1 = nhp_block_sum(t_sum, red);
        const double s2 = nhp_block_sum(t_sum2, red);
        if (tid == 0) {
            const size_t b = (size_t)bx + (size_t)by * gridDim.x;
            double *pp = EPI == EPI_GRAD ? g.partials2 : g.partials;
            pp[2 * b] = s1;
            pp[2 * b + 1] = s2;
        }
    }
    if (EPI == EPI_VB_Z || EPI == EPI_GRAD) {
        // deterministic column sums: lanes sharing r16 -> wave partial -> the two row-waves
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v = colp[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kk == 0) wcol[wave][j * 16 + r16] = v;
        }
        __syncthreads();
        if (tid < BN) {
            const int wn2 = tid >> 6, cl = tid & 63, col = n0 + tid;
            if (col < g.N) g.partials[(size_t)by * g.N + col] = wcol[wn2][cl] + wcol[2 + wn2][cl];
        }
    }
}

// ---- small kernels ----------------------------------------------------------------------------

// data (N x T, node fastest, Int64) -> dataT (T x N, t fastest, f64) + per-node event totals
__global__ __launch_bounds__(256) void k_disc_transpose(const int64_t *__restrict__ data, int N, int64_t T,
                                                        double *__restrict__ dataT, uint8_t *__restrict__ data8)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int n0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + tx;
        const int64_t t = t0 + r;
        if (n < N && t < T) tile[r][tx] = (double)data[(size_t)n + (size_t)t * N];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r;
        const int64_t t = t0 + tx;
        if (n < N && t < T) {
            dataT[(size_t)t + (size_t)n * T] = tile[tx][r];
            if (data8) data8[(size_t)t + (size_t)n * T] = (uint8_t)tile[tx][r];
        }
    }
}

// per-node Σ_t data[n,t]  and  Σ_t loggamma(data[n,t] + 1)  -> out[n], out[N + n]
__global__ __launch_bounds__(256) void k_disc_colstats(const double *__restrict__ dataT, int N, int64_t T,
                                                       double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    const int n = blockIdx.x;
    double s = 0.0, lg = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        const double v = dataT[(size_t)t + (size_t)n * T];
        s += v;
        if (v > 1.0) lg += lgamma(v + 1.0);
    }
    s = nhp_block_sum(s, red);
    lg = nhp_block_sum(lg, red);
    if (threadIdx.x == 0) { out[n] = s; out[N + n] = lg; }
}

// Ŝ[t,n,b] = max(0, Σ_{l=1..min(L,t)} data[n,t-l]·ϕ_b[l])   (0-based t; lag 0 excluded by the
// prepended 0.0 of the reference's conv kernel; direct form = the exact value its FFT approximates).
// A workgroup owns CONV_TB = 512 consecutive bins of one node, two per thread: the 512+L counts it needs sit in LDS (zeros
// before t = 0, which add exactly nothing) next to a bitmap of the NONZERO counts (one wave ballot per 64 bins), and the basis
// is transposed to ϕT[l][b] so one lag is a broadcast read.  A thread walks only the nonzero counts of its L-lag window
// (count data are sparse: BASELINE config 4 has 5 % occupied bins, 1.6 of 32 lags on average), most recent first, carrying
// CB basis sums at once.  Per basis the sum still runs l = 1..L in increasing order with separate multiply and add, and a
// skipped term is +0.0·ϕ -- which changes no bit of the running sum -- so the result is bit for bit the oracle's value.
// The two bins of a thread leave as ONE 16-byte non-temporal store per basis (8-byte stores ran at 1.8 TB/s; the 3.3 GB of
// Ŝ are re-read from HBM by the GEMMs whatever the cache policy).
#define CONV_CB 8
#ifndef CONV_PP
#define CONV_PP 4                    // pairs of bins per thread: a workgroup's prologue (tile, bitmap, basis) serves 2048 bins
#endif
#define CONV_TB (512 * CONV_PP)
typedef double nhp_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_disc_convolve(const double *__restrict__ dataT, const uint8_t *__restrict__ data8, int N, int64_t T,
                                                       const double *__restrict__ phi, int L, int B,
                                                       double *__restrict__ conv, double *__restrict__ colpart)
{
#pragma clang fp contract(off)
    extern __shared__ double csm[];
    double *tile = csm;                                  // [CONV_TB + L]: data[n, t0-L .. t0+CONV_TB-1]
    const int Bp = (B + CONV_CB - 1) / CONV_CB * CONV_CB;
    double *phiT = csm + CONV_TB + L;                    // [L][Bp], Bp = B rounded up to CONV_CB
    double *red = phiT + (size_t)L * Bp;                 // [4 waves][CONV_CB] column-sum staging
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(red + 4 * CONV_CB);   // [(CONV_TB + L) / 64 + 2]
    const int n = blockIdx.y, tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * CONV_TB;
    const double *d = dataT + (size_t)n * T;
    const uint8_t *d8 = data8 ? data8 + (size_t)n * T : nullptr;    // the counts in bytes (exact: they are integers <= 255)
    const int span = CONV_TB + L, nwords = (span + 63) / 64 + 1;
    for (int i = tid; i < nwords * 64; i += 256) {       // whole 64-bin words, so every ballot is a full word
        const int64_t tt = t0 - L + i;
#if defined(CONV_FILL) && CONV_FILL >= 2                 // timing experiment: no count reads either
        const double x = 0.0; (void)tt; (void)d;
#else
        const double x = (i < span && tt >= 0 && tt < T) ? (d8 ? (double)d8[tt] : d[tt]) : 0.0;
#endif
        if (i < span) tile[i] = x;
        const unsigned long long bal = __ballot(x != 0.0);
        if ((tid & 63) == 0) bits[i >> 6] = bal;
    }
    for (int i = tid; i < L * Bp; i += 256) {
        const int l = i / Bp, b = i % Bp;
        phiT[i] = b < B ? phi[l + (size_t)b * L] : 0.0;
    }
    __syncthreads();
    const unsigned long long lmask = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    const bool pair = (T & 1) == 0;                      // every (t, t+1) store then is 16-byte aligned
    for (int b0 = 0; b0 < B; b0 += CONV_CB) {
        double cs[CONV_CB];                              // this thread's share of Σ_t Ŝ[t, n, b] (the column sums the
#pragma unroll                                           // discrete adjacency sweep needs: no second pass over 3.3 GB)
        for (int q = 0; q < CONV_CB; ++q) cs[q] = 0.0;
        for (int pp = 0; pp < CONV_PP; ++pp) {
            const int o = 2 * (tid + 256 * pp);          // local bin of this pair's first output (a wave stores 1 KB runs)
            const int64_t t = t0 + o;
            if (t >= T) break;
            double s[2][CONV_CB];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // bit k of m <-> tile[o + j + k] = data[n, t + j - (L - k)]: the lags l = L - k of output j
                const int pos = o + j, sh = pos & 63;
                const unsigned long long lo = bits[pos >> 6], hi = bits[(pos >> 6) + 1];
                unsigned long long m = ((lo >> sh) | (sh ? hi << (64 - sh) : 0ull)) & lmask;
#pragma unroll
                for (int q = 0; q < CONV_CB; ++q) s[j][q] = 0.0;
#ifdef CONV_FILL                                         // timing experiment (tools/): the stores alone
                m = 0;
#endif
                while (m) {                              // increasing lag = decreasing bit: highest set bit first
                    const int k = 63 - __builtin_clzll(m);
                    m &= ~(1ull << k);
                    const double x = tile[o + j + k];
                    const double *ph = phiT + (size_t)(L - k - 1) * Bp + b0;
#pragma unroll
                    for (int q = 0; q < CONV_CB; ++q) s[j][q] = s[j][q] + x * ph[q];
                }
            }
#pragma unroll
            for (int q = 0; q < CONV_CB; ++q)
                if (b0 + q < B) {
                    double *dst = conv + (size_t)t + (size_t)n * T + (size_t)(b0 + q) * T * N;
                    const double v0 = s[0][q] > 0.0 ? s[0][q] : 0.0, v1 = (t + 1 < T && s[1][q] > 0.0) ? s[1][q] : 0.0;
                    cs[q] += v0 + v1;
                    if (pair) {
                        nhp_d2 v = {v0, v1};
#ifdef CONV_PLAIN
                        *reinterpret_cast<nhp_d2 *>(dst) = v;
#else
                        __builtin_nontemporal_store(v, reinterpret_cast<nhp_d2 *>(dst));
#endif
                    } else {
                        __builtin_nontemporal_store(v0, dst);
                        if (t + 1 < T) __builtin_nontemporal_store(v1, dst + 1);
                    }
                }
        }
        // per-workgroup column sums, waves in a fixed order (deterministic); k_disc_convsum_blocks adds the workgroups
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CONV_CB; ++q) {
            const double v = nhp_wave_sum(cs[q]);
            if ((tid & 63) == 0) red[(tid >> 6) * CONV_CB + q] = v;
        }
        __syncthreads();
        if (tid < CONV_CB && b0 + tid < B)
            colpart[((size_t)blockIdx.x * N + n) * B + b0 + tid] = (red[tid] + red[CONV_CB + tid]) + (red[2 * CONV_CB + tid] + red[3 * CONV_CB + tid]);
    }
}

// convsum[n + b·N] = Σ over the time blocks of colpart[blk][n][b], in block order
__global__ __launch_bounds__(256) void k_disc_convsum_blocks(const double *__restrict__ colpart, int nblk, int N, int B, double *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * B) return;
    const int n = i % N, b = i / N;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += colpart[((size_t)k * N + n) * B + b];
    out[i] = s;
}

// Dense form (every lag of every bin, 256 bins per workgroup, 8-byte stores): windows longer than the 64-lag bitmap.
__global__ __launch_bounds__(256) void k_disc_convolve_dense(const double *__restrict__ dataT, int N, int64_t T,
                                                       const double *__restrict__ phi, int L, int B,
                                                       double *__restrict__ conv)
{
#pragma clang fp contract(off)
    extern __shared__ double csm[];
    double *tile = csm;                                  // [256 + L]: data[n, t0-L .. t0+255]
    double *phiT = csm + 256 + L;                        // [L][Bp], Bp = B rounded up to CONV_CB
    const int n = blockIdx.y, tid = threadIdx.x;
    const int Bp = (B + CONV_CB - 1) / CONV_CB * CONV_CB;
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    const double *d = dataT + (size_t)n * T;
    for (int i = tid; i < 256 + L; i += 256) {
        const int64_t tt = t0 - L + i;
        tile[i] = (tt >= 0 && tt < T) ? d[tt] : 0.0;
    }
    for (int i = tid; i < L * Bp; i += 256) {
        const int l = i / Bp, b = i % Bp;
        phiT[i] = b < B ? phi[l + (size_t)b * L] : 0.0;
    }
    __syncthreads();
    const int64_t t = t0 + tid;
    for (int b0 = 0; b0 < B; b0 += CONV_CB) {
        double s[CONV_CB];
#pragma unroll
        for (int q = 0; q < CONV_CB; ++q) s[q] = 0.0;
        for (int l = 1; l <= L; ++l) {
            const double x = tile[tid + L - l];          // data[n, t - l]
            const double *ph = phiT + (size_t)(l - 1) * Bp + b0;
#pragma unroll
            for (int q = 0; q < CONV_CB; ++q) s[q] = s[q] + x * ph[q];
        }
        if (t < T) {
#pragma unroll
            for (int q = 0; q < CONV_CB; ++q)
                if (b0 + q < B) conv[(size_t)t + (size_t)n * T + (size_t)(b0 + q) * T * N] = s[q] > 0.0 ? s[q] : 0.0;
        }
    }
}

// E[k + c·K], k = p + b·N:  bump = ((a·)w·θ)·dt   (src/discrete.jl:381-385,511-516);  base[c] = λ0[c]·dt
__global__ __launch_bounds__(256) void k_disc_bump(int N, int B, double dt, const double *__restrict__ lambda0,
                                                   const double *__restrict__ W, const double *__restrict__ theta,
                                                   const double *__restrict__ A, double *__restrict__ E,
                                                   double *__restrict__ base, int cat_order)
{
#pragma clang fp contract(off)
    const size_t NN = (size_t)N * N, K = (size_t)N * B;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < NN * B) {
        const size_t b = i / NN, pc = i % NN, p = pc % N, c = pc / N;
        const double w = A ? A[pc] * W[pc] : W[pc];
        // cat_order: row q = p·B + b, the reference's parent-category order (src/parents.jl:108-112)
        E[(cat_order ? p * B + b : p + b * N) + c * K] = (w * theta[i]) * dt;
    }
    if (i < (size_t)N) base[i] = lambda0[i] * dt;
}

// [3P] SpecialFunctions.digamma, x > 0: recurrence to x >= 10 then the asymptotic series
__device__ __forceinline__ double nhp_digamma(double x)
{
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 +
                     f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + nhp_log(x) - 0.5 / x + t;
}

// VB factors from the OLD variational parameters (src/parents.jl:169-177):
// E[k + c·K] = exp(ψ(γv[p,c,b]) - ψ(Σ_b γv[p,c,·]) + ψ(κv[p,c]) - log νv[p,c]);  e0[c] = exp(ψ(αv) - log βv)
__global__ __launch_bounds__(256) void k_vb_factors(int N, int B, const double *__restrict__ alpha_v,
                                                    const double *__restrict__ beta_v,
                                                    const double *__restrict__ kappa_v,
                                                    const double *__restrict__ nu_v,
                                                    const double *__restrict__ gamma_v,
                                                    double *__restrict__ E, double *__restrict__ e0)
{
    const size_t NN = (size_t)N * N, K = (size_t)N * B;
    const size_t pc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pc < NN) {
        const size_t p = pc % N, c = pc / N;
        double gs = 0.0;
        for (int b = 0; b < B; ++b) gs += gamma_v[pc + (size_t)b * NN];
        const double elw = nhp_digamma(kappa_v[pc]) - nhp_log(nu_v[pc]);
        const double dgs = nhp_digamma(gs);
        for (int b = 0; b < B; ++b) {
            const double elt = nhp_digamma(gamma_v[pc + (size_t)b * NN]) - dgs;
            E[p + (size_t)b * N + c * K] = nhp_exp(elt + elw);
        }
    }
    if (pc < (size_t)N) e0[pc] = nhp_exp(nhp_digamma(alpha_v[pc]) - nhp_log(beta_v[pc]));
}

// After GEMM-2: Γ = E ⊙ Σ_z slab_z;  γv = γ + Γ,  κv = κ + Σ_b Γ,  νv[p,c] = ν + Σ_t data[p,t]
__global__ __launch_bounds__(256) void k_vb_finish(int N, int B, int n_slabs, const double *__restrict__ slabs,
                                                   const double *__restrict__ E, const double *__restrict__ colsum,
                                                   double kappa, double nu, double gamma,
                                                   double *__restrict__ kappa_v, double *__restrict__ nu_v,
                                                   double *__restrict__ gamma_v)
{
    const size_t NN = (size_t)N * N, K = (size_t)N * B;
    const size_t pc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pc >= NN) return;
    const size_t p = pc % N, c = pc / N;
    double ksum = 0.0;
    for (int b = 0; b < B; ++b) {
        const size_t o = p + (size_t)b * N + c * K;
        double s = 0.0;
        for (int z = 0; z < n_slabs; ++z) s += slabs[(size_t)z * K * N + o];
        const double G = E[o] * s;
        gamma_v[pc + (size_t)b * NN] = gamma + G;
        ksum += G;
    }
    kappa_v[pc] = kappa + ksum;
    nu_v[pc] = nu + colsum[p];
}

// αv[c] = α0 + e0[c]·Σ_t R[t,c];  βv[c] = 1/β0 + T·dt  (src/baselines.jl:444-452, D14 literal)
__global__ __launch_bounds__(256) void k_vb_baseline(int N, int row_blocks, const double *__restrict__ colpart,
                                                     const double *__restrict__ e0, double alpha0, double beta0,
                                                     double Tdt, double *__restrict__ alpha_v, double *__restrict__ beta_v)
{
    // 256 threads = 64 columns x 4 row-slices; fixed-order combine keeps the sum deterministic
    __shared__ double part[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < N)
        for (int r = sl; r < row_blocks; r += 4) s += colpart[(size_t)r * N + c];
    part[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < N) {
        alpha_v[c] = alpha0 + e0[c] * (((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl]);
        beta_v[c] = 1.0 / beta0 + Tdt;
    }
}

__global__ __launch_bounds__(256) void k_sum_pairs(const double *__restrict__ partials, int n, double lg_const,
                                                   double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { a += partials[2 * (size_t)i]; b += partials[2 * (size_t)i + 1]; }
    a = nhp_block_sum(a, red);
    b = nhp_block_sum(b, red);
    if (threadIdx.x == 0) *out = a - b - lg_const;
}

// ---- host side -----------------------------------------------------------------------------------

extern "C" nhp_status nhp_disc_basis(int32_t L, int32_t B, double dt, double *phi)
{
    // basis(impulse): src/impulses.jl:321-335 (the exponent parses as -d²/(4σ))
    if (L < 1 || B < 1 || !phi) return NHP_EINVAL;
    const double sigma = (double)L / (double)(B - 1);
    const double coef = ((-1.0 / 2.0) * (1.0 / sigma)) / 2.0;
    for (int b = 0; b < B; ++b) {
        const int len = (B < L) ? B + 2 : B, i = (B < L) ? b + 1 : b;
        const double tt = (double)i / (double)(len > 1 ? len - 1 : 1);
        const double mu = (1.0 - tt) * 1.0 + tt * (double)L;
        double s = 0.0;
        for (int l = 0; l < L; ++l) {
            const double d = (double)(l + 1) - mu;
            phi[l + (size_t)b * L] = exp(coef * (d * d));
            s += phi[l + (size_t)b * L];
        }
        for (int l = 0; l < L; ++l) phi[l + (size_t)b * L] /= (s * dt);
    }
    return NHP_OK;
}

// convsum[k] = Σ_t Ŝ[t, k]  (one workgroup per (p, b) column, fixed-order block reduction)
__global__ __launch_bounds__(256) void k_disc_convsum(const double *__restrict__ conv, int64_t T, double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    const double *col = conv + (size_t)blockIdx.x * (size_t)T;
    double s = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += 256) s += col[t];
    s = nhp_block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

extern "C" nhp_status nhp_disc_dataset_create(nhp_ctx *ctx, const int64_t *data, int32_t N, int64_t T,
                                              nhp_disc_dataset **out)
{
    if (!ctx || !out || !data || N < 1 || T < 1) return NHP_EINVAL;
    *out = nullptr;
    if (T >= ((int64_t)1 << 31) - BM) { nhp_set_error(ctx, "n_bins must be < 2^31"); return NHP_EINVAL; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    nhp_disc_dataset *ds = new nhp_disc_dataset();
    ds->ctx = ctx; ds->N = N; ds->T = T;
    const size_t NT = (size_t)N * (size_t)T;
    int64_t *d_raw = nullptr;
    hipStream_t st = ctx->stream;
    hipError_t e;
    if ((e = hipMalloc(&d_raw, 8 * NT)) != hipSuccess || (e = hipMalloc(&ds->d_dataT, 8 * NT)) != hipSuccess ||
        (e = hipMalloc(&ds->d_colsum, 8 * 2 * (size_t)N)) != hipSuccess) {
        nhp_set_error(ctx, "hipMalloc failed: %s", hipGetErrorString(e));
        if (d_raw) (void)hipFree(d_raw);
        nhp_disc_dataset_destroy(ds);
        return NHP_ENOMEM;
    }
    NHP_HIP(ctx, hipMemcpyAsync(d_raw, data, 8 * NT, hipMemcpyHostToDevice, st));
    int64_t vmax = 0, vmin = 0;
    {   // occupied bins by span of the time axis, within a span by (node, bin); data is N x T column-major
        const int64_t want = 2 * (int64_t)ctx->cu_count, least = (T + NHP_DA_SPAN - 1) / NHP_DA_SPAN;
        int64_t nsp = std::max<int64_t>(std::min<int64_t>(want, T), least);
        if (const char *es = getenv("NHP_DADJ_SPANS")) { const int64_t v = atoll(es); if (v >= least && v <= T) nsp = v; }
        std::vector<int32_t> ot, oc, off((size_t)nsp + 1, 0), spt((size_t)nsp + 1, 0);
        std::vector<double> os;
        std::vector<int32_t> st_, sc_, cnt((size_t)N + 1);
        std::vector<double> ss_;
        int64_t nocc = 0;
        for (int64_t k = 0; k < nsp; ++k) {
            const int64_t ta = k * T / nsp, tb = (k + 1) * T / nsp;              // (tb - ta <= ceil(T / nsp) <= NHP_DA_SPAN)
            spt[(size_t)k] = (int32_t)ta;
            off[(size_t)k] = (int32_t)ot.size();
            st_.clear(); sc_.clear(); ss_.clear();
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int64_t t = ta; t < tb; ++t)
                for (int32_t n = 0; n < N; ++n) {
                    const int64_t v = data[(size_t)n + (size_t)t * N];
                    if (v > 0) { st_.push_back((int32_t)t); sc_.push_back(n); ss_.push_back((double)v); ++cnt[(size_t)n + 1]; }
                    if (v > vmax) vmax = v;
                    if (v < vmin) vmin = v;
                }
            for (int32_t n = 0; n < N; ++n) cnt[(size_t)n + 1] += cnt[(size_t)n];
            const size_t base = ot.size(), m = st_.size(), mp = (m + 3) / 4 * 4;
            ot.resize(base + mp, (int32_t)ta); oc.resize(base + mp, 0); os.resize(base + mp, 0.0);      // (padding: bin ta, node 0, count 0)
            for (size_t i = 0; i < m; ++i) {                                       // stable by node: bins stay ascending
                const size_t d = base + (size_t)cnt[(size_t)sc_[i]]++;
                ot[d] = st_[i]; oc[d] = sc_[i]; os[d] = ss_[i];
            }
            nocc += (int64_t)m;
            ds->da_max_entries = std::max<int32_t>(ds->da_max_entries, (int32_t)mp);
            if (ot.size() >= ((size_t)1 << 31)) break;
        }
        off[(size_t)nsp] = (int32_t)ot.size();
        spt[(size_t)nsp] = (int32_t)T;
        ds->nocc = nocc;
        ds->nocc_pad = (int64_t)ot.size();
        ds->da_nspans = (int32_t)nsp;
        if (ot.size() >= ((size_t)1 << 31)) { (void)hipFree(d_raw); nhp_set_error(ctx, "too many occupied bins"); nhp_disc_dataset_destroy(ds); return NHP_ENOTIMPL; }
        const size_t no = ot.size() ? ot.size() : 4;
        if (hipMalloc(&ds->d_occ_t, 4 * no) != hipSuccess || hipMalloc(&ds->d_occ_c, 4 * no) != hipSuccess ||
            hipMalloc(&ds->d_occ_s, 8 * no) != hipSuccess || hipMalloc(&ds->d_occ_off, 4 * off.size()) != hipSuccess ||
            hipMalloc(&ds->d_span_t, 4 * spt.size()) != hipSuccess) {
            (void)hipFree(d_raw); nhp_set_error(ctx, "out of device memory (occupied-bin list)"); nhp_disc_dataset_destroy(ds); return NHP_ENOMEM;
        }
        if (!ot.empty()) {
            NHP_HIP(ctx, hipMemcpy(ds->d_occ_t, ot.data(), 4 * ot.size(), hipMemcpyHostToDevice));
            NHP_HIP(ctx, hipMemcpy(ds->d_occ_c, oc.data(), 4 * oc.size(), hipMemcpyHostToDevice));
            NHP_HIP(ctx, hipMemcpy(ds->d_occ_s, os.data(), 8 * os.size(), hipMemcpyHostToDevice));
        }
        NHP_HIP(ctx, hipMemcpy(ds->d_occ_off, off.data(), 4 * off.size(), hipMemcpyHostToDevice));
        NHP_HIP(ctx, hipMemcpy(ds->d_span_t, spt.data(), 4 * spt.size(), hipMemcpyHostToDevice));
        // the adjacency sweep's entry word: node (16 bits) | count (8 bits) | bin % 256 (a span is at most 256 bins)
        static_assert(NHP_DA_SPAN == 256, "the packed entry keeps 8 bits of the bin");
        if (N <= 65536 && vmax < 256 && !ot.empty() && hipMalloc((void **)&ds->d_occ_pack, 4 * no) == hipSuccess) {
            std::vector<uint32_t> pk(ot.size());
            for (size_t i = 0; i < ot.size(); ++i)
                pk[i] = ((uint32_t)oc[i] << 16) | ((uint32_t)os[i] << 8) | (uint32_t)(ot[i] % NHP_DA_SPAN);
            NHP_HIP(ctx, hipMemcpy(ds->d_occ_pack, pk.data(), 4 * pk.size(), hipMemcpyHostToDevice));
        } else {
            (void)hipGetLastError();
            ds->d_occ_pack = nullptr;
        }
    }
    // counts that fit a byte are kept in bytes too: the convolution then reads 1/8 of the bytes next to its 3.3 GB of stores
    if (vmin >= 0 && vmax <= 255 && hipMalloc((void **)&ds->d_data8, NT) != hipSuccess) ds->d_data8 = nullptr;
    dim3 tg((unsigned)((T + 31) / 32), (unsigned)((N + 31) / 32));
    hipLaunchKernelGGL(k_disc_transpose, tg, dim3(256), 0, st, d_raw, N, T, ds->d_dataT, ds->d_data8);
    hipLaunchKernelGGL(k_disc_colstats, dim3((unsigned)N), dim3(256), 0, st, ds->d_dataT, N, T, ds->d_colsum);
    NHP_HIP(ctx, hipGetLastError());
    std::vector<double> h(2 * (size_t)N);
    NHP_HIP(ctx, hipMemcpyAsync(h.data(), ds->d_colsum, 8 * 2 * (size_t)N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    (void)hipFree(d_raw);
    ds->lgamma_sum = 0.0;
    for (int n = 0; n < N; ++n) {
        ds->lgamma_sum += h[(size_t)N + n];
        if (h[n] < 0.0) { nhp_set_error(ctx, "counts must be non-negative"); nhp_disc_dataset_destroy(ds); return NHP_EDOMAIN; }
    }
    *out = ds;
    return NHP_OK;
}

extern "C" void nhp_disc_dataset_destroy(nhp_disc_dataset *ds)
{
    if (!ds) return;
    (void)hipSetDevice(ds->ctx->device);
    (void)hipStreamSynchronize(ds->ctx->stream);
    (void)hipFree(ds->d_dataT); (void)hipFree(ds->d_data8); (void)hipFree(ds->d_conv); (void)hipFree(ds->d_colsum);
    (void)hipFree(ds->d_occ_t); (void)hipFree(ds->d_occ_c); (void)hipFree(ds->d_occ_s); (void)hipFree(ds->d_occ_off); (void)hipFree(ds->d_occ_pack); (void)hipFree(ds->d_span_t);
    (void)hipFree(ds->d_convsum); (void)hipFree(ds->d_baseT); (void)hipFree(ds->d_base_counts);
    delete ds;
}

extern "C" nhp_status nhp_disc_convolve(nhp_ctx *ctx, nhp_disc_dataset *ds, const double *phi, int32_t L,
                                        int32_t B, double *out)
{
    if (!ctx || !ds || !phi || L < 1 || B < 1) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t TNB = (size_t)ds->T * ds->N * B;
    hipStream_t st = ctx->stream;
    if (ds->B != B || !ds->d_conv) {
        NHP_HIP(ctx, hipStreamSynchronize(st));
        if (ds->d_conv) (void)hipFree(ds->d_conv);
        ds->d_conv = nullptr;
        if (hipMalloc(&ds->d_conv, 8 * TNB) != hipSuccess) { nhp_set_error(ctx, "out of device memory for the %zu-byte convolution", 8 * TNB); return NHP_ENOMEM; }
    }
    ds->B = B; ds->L = L;
    const size_t part = (size_t)((ds->T + CONV_TB - 1) / CONV_TB) * ds->N * B;       // per-workgroup column sums (sparse kernel)
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * ((size_t)L * B + part)));
    double *d_phi = (double *)ctx->d_scratch, *d_part = d_phi + (size_t)L * B;
    NHP_HIP(ctx, hipMemcpyAsync(d_phi, phi, 8 * (size_t)L * B, hipMemcpyHostToDevice, st));
    // sparse walk (nonzero counts of the lag window only) for count data as sparse as the models assume; a dense matrix
    // (more than ~1 in 8 bins occupied) or more than 64 lags takes the dense kernel
    const bool sparse = L <= 64 && !getenv("NHP_CONV_DENSE") && (double)ds->nocc < 0.125 * (double)ds->N * (double)ds->T;
    const int tb = sparse ? CONV_TB : 256;
    dim3 grid((unsigned)((ds->T + tb - 1) / tb), (unsigned)ds->N);
    const size_t lds_conv = 8 * ((size_t)tb + L + (size_t)L * ((B + CONV_CB - 1) / CONV_CB * CONV_CB) + (sparse ? 4 * CONV_CB + (CONV_TB + L + 63) / 64 + 2 : 0));
    if (lds_conv > 64 * 1024) { nhp_set_error(ctx, "convolve: nlags * nbasis = %d * %d exceeds the LDS budget", L, B); return NHP_ENOTIMPL; }
    if (ds->d_convsum) { (void)hipFree(ds->d_convsum); ds->d_convsum = nullptr; }
    if (hipMalloc(&ds->d_convsum, 8 * (size_t)ds->N * B) != hipSuccess) { nhp_set_error(ctx, "out of device memory"); return NHP_ENOMEM; }
    if (sparse) {
        hipLaunchKernelGGL(k_disc_convolve, grid, dim3(256), lds_conv, st, ds->d_dataT, (const uint8_t *)ds->d_data8, ds->N, ds->T, d_phi, L, B, ds->d_conv, d_part);
        NHP_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_disc_convsum_blocks, dim3((unsigned)(((size_t)ds->N * B + 255) / 256)), dim3(256), 0, st, d_part, (int)grid.x, ds->N, B, ds->d_convsum);
    } else {
        hipLaunchKernelGGL(k_disc_convolve_dense, grid, dim3(256), lds_conv, st, ds->d_dataT, ds->N, ds->T, d_phi, L, B, ds->d_conv);
        NHP_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_disc_convsum, dim3((unsigned)((size_t)ds->N * B)), dim3(256), 0, st, ds->d_conv, ds->T, ds->d_convsum);
    }
    NHP_HIP(ctx, hipGetLastError());
    if (out) NHP_HIP(ctx, hipMemcpyAsync(out, ds->d_conv, 8 * TNB, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

// Rows per output tile of the m-contiguous GEMM-1 (M = T bins): the workgroups of a launch run in rounds of 2 per CU,
// and the last round is rarely full -- 3128 tiles of 128 rows on 512 slots are 6.1 rounds, i.e. 7; the same matrix in
// 160-row tiles is 2500 tiles = 4.9 rounds, i.e. 5 x 1.25: 11 % less.  Pick the cheaper of the two (NHP_GEMM_BM forces).
static int gemm1_tile_m(int64_t M, int N, int cu_count)
{
    const char *env = getenv("NHP_GEMM_BM");                     // read per call: tests force both tile heights
    const int forced = env ? atoi(env) : 0;
    if (forced == 128 || forced == 160) return forced;
    const int64_t slots = 2 * (int64_t)(cu_count > 0 ? cu_count : 256), nt = (N + BN - 1) / BN;
    auto cost = [&](int bm) { const int64_t tiles = ((M + bm - 1) / bm) * nt; return ((tiles + slots - 1) / slots) * bm; };
    return cost(160) < cost(128) ? 160 : 128;
}

template <bool AMC, int EPI>
static void launch_gemm(const gemm_args &g, int splits, hipStream_t st, int bm = BM)
{
    // whole tiles only (and every reduction chunk whole BK-tiles): the branch-free, instruction-scheduled main loop
    const int tbm = AMC && bm == 160 ? 160 : BM;
    const bool whole = g.M % tbm == 0 && g.N % BN == 0 && g.K % BK == 0 && g.k_chunk % BK == 0 && g.K > 0 && !getenv("NHP_GEMM_PLAIN");
    auto go = [&](auto kernel, int rows) {
        dim3 grid((unsigned)((g.N + BN - 1) / BN), (unsigned)((g.M + rows - 1) / rows), (unsigned)splits);
        const size_t lds = 8 * 2 * ((AMC ? BK * (rows + 16) : BM * KC_LD) + BN * KC_LD);   // two stages (>= the epilogue's NHP_WAVES * 65 doubles)
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, g);
    };
    if (AMC && bm == 160) {
        if (whole) go(k_gemm_f64<AMC, EPI, (AMC ? 160 : BM), true>, 160);
        else go(k_gemm_f64<AMC, EPI, (AMC ? 160 : BM), false>, 160);
        return;
    }
    if (whole) go(k_gemm_f64<AMC, EPI, BM, true>, BM);
    else go(k_gemm_f64<AMC, EPI, BM, false>, BM);
}

// uploads the model pieces, builds E and base on the device; returns pointers into scratch
nhp_status nhp_disc_stage_bump(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0, const double *W,
                               const double *theta, const double *A, double dt, double **E, double **base, size_t extra,
                               double **extra_ptr, int cat_order)
{
    if (!ds->d_conv) { nhp_set_error(ctx, "convolve(process, data) must run before intensity / loglikelihood"); return NHP_EINVAL; }
    if (!W || !theta) return NHP_EINVAL;
    if (!lambda0 && !ds->d_baseT) { nhp_set_error(ctx, "lambda0 is NULL and the dataset has no LGCP baseline (nhp_disc_set_lgcp_baseline)"); return NHP_EINVAL; }
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B, K = N * B;
    const size_t need = 8 * (K * N + N + N + NN + NN * B + NN + extra);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, need));
    double *p = (double *)ctx->d_scratch;
    double *dE = p; p += K * N;
    double *dbase = p; p += N;
    double *dl0 = p; p += N;
    double *dW = p; p += NN;
    double *dth = p; p += NN * B;
    double *dA = p; p += NN;
    *extra_ptr = p;
    hipStream_t st = ctx->stream;
    if (lambda0) NHP_HIP(ctx, hipMemcpyAsync(dl0, lambda0, 8 * N, hipMemcpyHostToDevice, st));
    else NHP_HIP(ctx, hipMemsetAsync(dl0, 0, 8 * N, st));          // per-bin baseline comes from ds->d_baseT
    NHP_HIP(ctx, hipMemcpyAsync(dW, W, 8 * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dth, theta, 8 * NN * B, hipMemcpyHostToDevice, st));
    if (A) NHP_HIP(ctx, hipMemcpyAsync(dA, A, 8 * NN, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_disc_bump, dim3((unsigned)((NN * B + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, dt, dl0, dW,
                       dth, A ? dA : nullptr, dE, dbase, cat_order);
    NHP_HIP(ctx, hipGetLastError());
    *E = dE; *base = dbase;
    return NHP_OK;
}

// λ = base ⊕ G·E into dlam [T x N] (GEMM-1 with the plain epilogue); shared with the adjacency sweep
nhp_status nhp_disc_launch_intensity(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *E, const double *base,
                                     bool per_bin_baseline, double *dlam)
{
    gemm_args g{};
    g.A = ds->d_conv; g.lda = (size_t)ds->T; g.B = E; g.ldb = (size_t)ds->N * ds->B;
    g.M = (int)ds->T; g.N = ds->N; g.K = ds->N * ds->B; g.k_chunk = g.K;
    g.base = base; g.baseT = per_bin_baseline ? ds->d_baseT : nullptr; g.out = dlam;
    launch_gemm<true, EPI_INTENSITY>(g, 1, ctx->stream, gemm1_tile_m(g.M, g.N, ctx->cu_count));
    NHP_HIP(ctx, hipGetLastError());
    return NHP_OK;
}

extern "C" nhp_status nhp_disc_intensity(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                         const double *W, const double *theta, const double *A, double dt,
                                         double *lam)
{
    if (!ctx || !ds || !lam) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t TN = (size_t)ds->T * ds->N;
    double *E, *base, *dlam;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E, &base, TN, &dlam));
    gemm_args g{};
    g.A = ds->d_conv; g.lda = (size_t)ds->T; g.B = E; g.ldb = (size_t)ds->N * ds->B;
    g.M = (int)ds->T; g.N = ds->N; g.K = ds->N * ds->B; g.k_chunk = g.K;
    g.base = base; g.baseT = lambda0 ? nullptr : ds->d_baseT; g.out = dlam;
    launch_gemm<true, EPI_INTENSITY>(g, 1, ctx->stream, gemm1_tile_m(g.M, g.N, ctx->cu_count));
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(lam, dlam, 8 * TN, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NHP_OK;
}

extern "C" nhp_status nhp_disc_loglik(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                      const double *W, const double *theta, const double *A, double dt, double *ll)
{
    if (!ctx || !ds || !ll) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    double *E, *base, *unused;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E, &base, 0, &unused));
    gemm_args g{};
    g.A = ds->d_conv; g.lda = (size_t)ds->T; g.B = E; g.ldb = (size_t)ds->N * ds->B;
    g.M = (int)ds->T; g.N = ds->N; g.K = ds->N * ds->B; g.k_chunk = g.K;
    g.base = base; g.baseT = lambda0 ? nullptr : ds->d_baseT; g.dataT = ds->d_dataT;
    const int bm = gemm1_tile_m(g.M, g.N, ctx->cu_count);
    const int blocks = ((g.M + bm - 1) / bm) * ((g.N + BN - 1) / BN);
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)blocks));
    g.partials = ctx->d_partials;
    launch_gemm<true, EPI_LOGLIK>(g, 1, ctx->stream, bm);
    NHP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_sum_pairs, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, blocks, ds->lgamma_sum, ctx->d_results);
    NHP_HIP(ctx, hipGetLastError());
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}

// ---- DiscreteLogGaussianCoxProcess baseline (reference src/baselines.jl:461-609).  intensity(p, ts) is the
// piecewise-linear interpolation of (x, λ[:, n]·dt); the process intensity evaluates it at ts = 1..T
// (src/discrete.jl:117), the baseline's own likelihood at range(p) = x[1] : dt : x[end]-dt (:553-584).
__device__ __forceinline__ void disc_interp_cell(const double *x, int G, double t, int *lo, double *wlo, double *whi)
{
    // src/utils/interpolation.jl:26-35: x0 in [x[i], x[i+1]) -> cell i; x0 >= x[end] -> last value
    int a = 0, b = G - 1;
    if (!(t < x[b])) { *lo = b; *wlo = 1.0; *whi = 0.0; return; }
    while (b - a > 1) { const int mid = (a + b) >> 1; if (t >= x[mid]) a = mid; else b = mid; }
    const double w = x[a + 1] - x[a];
    *lo = a; *wlo = (x[a + 1] - t) / w; *whi = (t - x[a]) / w;
}

// baseT[t + T·c] = interp(x, lam[:, c]·dt)(t + 1)
__global__ __launch_bounds__(256) void k_disc_base_interp(const double *__restrict__ x, int G, const double *__restrict__ lam,
                                                          double dt, int64_t T, double *__restrict__ baseT)
{
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= T) return;
    const double time = (double)(t + 1);
    const double *y = lam + (size_t)c * G;
    int lo; double wlo, whi;
    disc_interp_cell(x, G, time, &lo, &wlo, &whi);
    double v;
    if (lo == G - 1) v = y[G - 1] * dt;
    else v = ((y[lo + 1] * dt) * (time - x[lo]) + (y[lo] * dt) * (x[lo + 1] - time)) / (x[lo + 1] - x[lo]);
    baseT[(size_t)t + (size_t)T * c] = v;
}

// grad[g + G·c] += dt Σ_t (R[t,c] - 1) w_g(t + 1)
__global__ __launch_bounds__(256) void k_disc_grad_lgcp(const double *__restrict__ R, int64_t T, int G, const double *__restrict__ x,
                                                        double dt, double *__restrict__ grad)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= T) return;
    const double r = (R[(size_t)t + (size_t)T * c] - 1.0) * dt;
    int lo; double wlo, whi;
    disc_interp_cell(x, G, (double)(t + 1), &lo, &wlo, &whi);
    atomicAdd(&grad[(size_t)c * G + lo], r * wlo);
    if (whi != 0.0) atomicAdd(&grad[(size_t)c * G + lo + 1], r * whi);
}

// ll[c] = Σ_t log pdf(Poisson(λ_t), s0[t,c]),  λ_t = interp(x, cand[:, c]·dt)(x[0] + t·dt)   (src/baselines.jl:571-584)
__global__ __launch_bounds__(256) void k_disc_lgcp_ll(const int *__restrict__ s0, int64_t T, int G, const double *__restrict__ x,
                                                      const double *__restrict__ cand, double dt, double *__restrict__ ll)
{
#pragma clang fp contract(off)
    __shared__ double red[NHP_WAVES];
    const int c = blockIdx.x;
    const double *y = cand + (size_t)c * G;
    double acc = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        const double time = x[0] + (double)t * dt;
        int lo; double wlo, whi;
        disc_interp_cell(x, G, time, &lo, &wlo, &whi);
        double lam;
        if (lo == G - 1) lam = y[G - 1] * dt;
        else lam = ((y[lo + 1] * dt) * (time - x[lo]) + (y[lo] * dt) * (x[lo + 1] - time)) / (x[lo + 1] - x[lo]);
        const double s = (double)s0[(size_t)t + (size_t)T * c];
        acc += (s == 0.0 ? 0.0 : s * nhp_log(lam)) - lam - lgamma(s + 1.0);
    }
    acc = nhp_block_sum(acc, red);
    if (threadIdx.x == 0) ll[c] = acc;
}

// ---- gradient of the discrete log-likelihood in the reference's mle! parameters [λ0; η = W∘θ]
// (params / params! src/discrete.jl:174-201; the reference hands Optim no gradient -- its d_loglikelihood
// :298-314 is commented out of mle! -- so one gradient costs it 2(N + N²B) objective calls):
//     ∂ll/∂λ0[c] = dt Σ_t (s_tc/λ_tc - 1),      ∂ll/∂η[p,c,b] = dt Σ_t (s_tc/λ_tc - 1) Ŝ[t,p,b]
// i.e. with R = data/Z from GEMM-1:  dt (colsum R - T)  and  dt (Gᵀ·R - Σ_t Ŝ) -- the VB step's two GEMMs
// with a different last kernel.
__global__ __launch_bounds__(256) void k_disc_grad_finish(int N, int B, int splits, int row_blocks, double dt, double Tbins,
                                                          const double *__restrict__ slabs, const double *__restrict__ colp,
                                                          const double *__restrict__ convsum, double *__restrict__ grad)
{
    const size_t NN = (size_t)N * N, K = (size_t)N * B;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < NN * B) {
        const size_t b = i / NN, pc = i % NN, p = pc % N, c = pc / N, k = p + b * N;
        double g = 0.0;
        for (int z = 0; z < splits; ++z) g += slabs[(size_t)z * K * N + k + c * K];      // fixed order
        grad[N + i] = dt * (g - convsum[k]);
    }
    if (i < (size_t)N) {
        double r = 0.0;
        for (int rb = 0; rb < row_blocks; ++rb) r += colp[(size_t)rb * N + i];
        grad[i] = dt * (r - Tbins);
    }
}

// Sizes of one (log-likelihood, gradient) evaluation and its scratch behind the bump table
struct disc_grad_plan {
    size_t N, NN, B, K, T, G, nbase;
    int bm, row_blocks, splits, k_chunk;
    size_t extra() const { return T * N + (size_t)row_blocks * N + (size_t)splits * K * N + nbase + NN * B + G; }
};

static disc_grad_plan disc_grad_sizes(nhp_ctx *ctx, const nhp_disc_dataset *ds, bool homogeneous)
{
    disc_grad_plan q{};
    q.N = (size_t)ds->N; q.NN = q.N * q.N; q.B = (size_t)ds->B; q.K = q.N * q.B; q.T = (size_t)ds->T;
    q.G = homogeneous ? 0 : ds->h_grid_x.size();
    q.nbase = homogeneous ? q.N : q.G * q.N;                     // params(baseline): λ or vec(λ) (G x N)
    q.bm = gemm1_tile_m((int64_t)q.T, (int)q.N, ctx->cu_count);
    q.row_blocks = (int)((q.T + q.bm - 1) / q.bm);
    const int tiles2 = (int)(((q.K + BM - 1) / BM) * ((q.N + BN - 1) / BN));
    int splits = (2 * ctx->cu_count + tiles2 - 1) / tiles2;
    splits = std::max(1, std::min(splits, (int)((q.T + 4 * BK - 1) / (4 * BK))));
    int k_chunk = (int)((q.T + splits - 1) / splits);
    k_chunk = ((k_chunk + BK - 1) / BK) * BK;
    q.k_chunk = k_chunk;
    q.splits = (int)((q.T + k_chunk - 1) / k_chunk);
    return q;
}

// log-likelihood -> ctx->d_results[0], gradient in [params(baseline); vec(W .* θ)] order -> *dgrad_out (inside `x`, the
// scratch behind the bump table E / base); asynchronous
static nhp_status disc_grad_enqueue(nhp_ctx *ctx, const nhp_disc_dataset *ds, const disc_grad_plan &q, bool homogeneous, const double *E,
                                    const double *base, double *x, double dt, double **dgrad_out)
{
    const size_t N = q.N, NN = q.NN, B = q.B, K = q.K, T = q.T, G = q.G, nbase = q.nbase;
    double *dR = x; x += T * N;
    double *dcolp = x; x += (size_t)q.row_blocks * N;
    double *dslab = x; x += (size_t)q.splits * K * N;
    double *dgrad = x;
    hipStream_t st = ctx->stream;
    // GEMM-1 once, with both epilogues: the Poisson log-likelihood partials AND R = data / Z with its column sums
    gemm_args g{};
    g.A = ds->d_conv; g.lda = T; g.B = E; g.ldb = K; g.M = (int)T; g.N = (int)N; g.K = (int)K; g.k_chunk = (int)K;
    g.base = base; g.baseT = homogeneous ? nullptr : ds->d_baseT; g.dataT = ds->d_dataT;
    const int blocks = ((g.M + q.bm - 1) / q.bm) * ((g.N + BN - 1) / BN);
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)blocks));
    g.partials2 = ctx->d_partials;
    g.out = dR; g.partials = dcolp;
    launch_gemm<true, EPI_GRAD>(g, 1, st, q.bm);
    hipLaunchKernelGGL(k_sum_pairs, dim3(1), dim3(256), 0, st, ctx->d_partials, blocks, ds->lgamma_sum, ctx->d_results);
    // then Gᵀ·R in T-slabs
    gemm_args g2{};
    g2.A = ds->d_conv; g2.lda = T; g2.B = dR; g2.ldb = T; g2.M = (int)K; g2.N = (int)N; g2.K = (int)T; g2.k_chunk = q.k_chunk;
    g2.out = dslab;
    launch_gemm<false, EPI_SLAB>(g2, q.splits, st);
    hipLaunchKernelGGL(k_disc_grad_finish, dim3((unsigned)((NN * B + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, q.splits, q.row_blocks,
                       dt, (double)T, dslab, dcolp, ds->d_convsum, dgrad + (nbase - N));
    if (!homogeneous) {
        // LGCP: ∂λ_base[t,c]/∂λgrid[g,c] = dt·w_g(t) (interpolation weights at bin time t+1), so the baseline block
        // is dt Σ_t (R - 1)[t,c] w_g -- it overwrites the N homogeneous entries k_disc_grad_finish left at its end
        double *dgx = dgrad + nbase + NN * B;
        NHP_HIP(ctx, hipMemcpyAsync(dgx, ds->h_grid_x.data(), 8 * G, hipMemcpyHostToDevice, st));
        NHP_HIP(ctx, hipMemsetAsync(dgrad, 0, 8 * nbase, st));
        hipLaunchKernelGGL(k_disc_grad_lgcp, dim3((unsigned)((T + 255) / 256), (unsigned)N), dim3(256), 0, st, dR, (int64_t)T, (int)G,
                           dgx, dt, dgrad);
    }
    NHP_HIP(ctx, hipGetLastError());
    *dgrad_out = dgrad;
    return NHP_OK;
}

extern "C" nhp_status nhp_disc_loglik_grad(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                           const double *W, const double *theta, double dt, double *ll, double *grad,
                                           int64_t grad_len)
{
    if (!ctx || !ds || !ll || !grad) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const disc_grad_plan q = disc_grad_sizes(ctx, ds, lambda0 != nullptr);
    if ((size_t)grad_len != q.nbase + q.NN * q.B) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    if (!ds->d_convsum) { nhp_set_error(ctx, "convolve(process, data) must run before the gradient"); return NHP_EINVAL; }
    double *E, *base, *x, *dgrad;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, nullptr, dt, &E, &base, q.extra(), &x));
    NHP_TRY(disc_grad_enqueue(ctx, ds, q, lambda0 != nullptr, E, base, x, dt, &dgrad));
    NHP_HIP(ctx, hipMemcpyAsync(grad, dgrad, 8 * (q.nbase + q.NN * q.B), hipMemcpyDeviceToHost, ctx->stream));
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}

// ---- mle!(process::DiscreteStandardHawkesProcess, data) with the optimizer's state on the device (src/discrete.jl:211-296) ----
// x = params(process) = [λ0; vec(W .* θ)] (src/discrete.jl:178-182).  params!(process, x) (:195-203) splits the second block
// into W = Σ_b x[p,c,b] and θ = x ./ W, and the objective's bump is (W·θ)·dt (:381-385): the same three roundings here, from
// the DEVICE vector.  E[k + c·K], k = p + b·N; base[c] = λ0[c]·dt.
__global__ __launch_bounds__(256) void k_disc_bump_from_x(int N, int B, double dt, const double *__restrict__ x, double *__restrict__ E,
                                                          double *__restrict__ base)
{
#pragma clang fp contract(off)
    const size_t NN = (size_t)N * N, K = (size_t)N * B;
    const size_t pc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pc < NN) {
        const size_t p = pc % N, c = pc / N;
        const double *xw = x + N;
        double w = 0.0;
        for (int b = 0; b < B; ++b) w += xw[pc + (size_t)b * NN];
        for (int b = 0; b < B; ++b) {
            const double th = xw[pc + (size_t)b * NN] / w;
            E[p + (size_t)b * N + c * K] = (w * th) * dt;
        }
    }
    if (pc < (size_t)N) base[pc] = x[pc] * dt;
}

#include "nhp_lbfgs.h"

extern "C" nhp_status nhp_disc_mle_run(nhp_ctx *ctx, const nhp_disc_dataset *ds, double dt, double lower, double upper, double f_abstol,
                                       int32_t max_steps, double *x, int64_t P, double *loss, int32_t *steps_out, int32_t *converged_out,
                                       int32_t *evals_out)
{
    if (!ctx || !ds || !x || !loss || !steps_out || !converged_out) return NHP_EINVAL;
    if (!(lower < upper) || max_steps < 0 || !(dt > 0.0)) return NHP_EDOMAIN;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    if (!ds->d_conv || !ds->d_convsum) { nhp_set_error(ctx, "convolve(process, data) must run before mle!"); return NHP_EINVAL; }
    const disc_grad_plan q = disc_grad_sizes(ctx, ds, true);
    if ((size_t)P != q.nbase + q.NN * q.B) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    // scratch: E | base | the evaluation's buffers (nhp_disc_stage_bump's layout without its host-side copies)
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (q.K * q.N + q.N + q.extra())));
    auto eval = [&](const double *d_x, double *d_g, bool /*commit*/) -> nhp_status {
        hipStream_t st = ctx->stream;
        double *E = (double *)ctx->d_scratch, *base = E + q.K * q.N, *xs = base + q.N, *dgrad = nullptr;
        hipLaunchKernelGGL(k_disc_bump_from_x, dim3((unsigned)((q.NN + 255) / 256)), dim3(256), 0, st, (int)q.N, (int)q.B, dt, d_x, E, base);
        NHP_TRY(disc_grad_enqueue(ctx, ds, q, true, E, base, xs, dt, &dgrad));
        hipLaunchKernelGGL(k_mle_neg, dim3((unsigned)std::min<int64_t>(2048, (P + 255) / 256)), dim3(256), 0, st, d_g, (const double *)dgrad, P);
        NHP_HIP(ctx, hipGetLastError());
        return NHP_OK;
    };
    return nhp_lbfgs_box(ctx, P, lower, upper, f_abstol, max_steps, eval, x, loss, steps_out, converged_out, evals_out);
}

// (nhp_disc_vb_run is declared in include/nhp.h)
extern "C" nhp_status nhp_disc_vb_step(nhp_ctx *ctx, const nhp_disc_dataset *ds, double dt,
                                       double alpha0, double beta0, double kappa, double nu, double gamma,
                                       double *alpha_v, double *beta_v, double *kappa_v, double *nu_v, double *gamma_v)
{
    return nhp_disc_vb_run(ctx, ds, dt, alpha0, beta0, kappa, nu, gamma, 1, alpha_v, beta_v, kappa_v, nu_v, gamma_v);
}

extern "C" nhp_status nhp_disc_vb_run(nhp_ctx *ctx, const nhp_disc_dataset *ds, double dt,
                                      double alpha0, double beta0, double kappa, double nu, double gamma, int32_t n_steps,
                                      double *alpha_v, double *beta_v, double *kappa_v, double *nu_v, double *gamma_v)
{
    if (n_steps < 1) return NHP_EINVAL;
    if (!ctx || !ds || !alpha_v || !beta_v || !kappa_v || !nu_v || !gamma_v) return NHP_EINVAL;
    if (!ds->d_conv) { nhp_set_error(ctx, "convolve(process, data) must run before update!"); return NHP_EINVAL; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B, K = N * B, T = (size_t)ds->T;
    const int bm = gemm1_tile_m((int64_t)T, (int)N, ctx->cu_count);
    const int row_blocks = (int)((T + bm - 1) / bm);
    // split the T-long reduction of GEMM-2 so that the grid fills the chip
    const int tiles2 = (int)(((K + BM - 1) / BM) * ((N + BN - 1) / BN));
    int splits = (2 * ctx->cu_count + tiles2 - 1) / tiles2;
    splits = std::max(1, std::min(splits, (int)((T + 4 * BK - 1) / (4 * BK))));
    int k_chunk = (int)((T + splits - 1) / splits);
    k_chunk = ((k_chunk + BK - 1) / BK) * BK;
    splits = (int)((T + k_chunk - 1) / k_chunk);
    const size_t need = 8 * (K * N + N + N + N + NN + NN + NN * B + T * N + (size_t)row_blocks * N + (size_t)splits * K * N);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, need));
    double *p = (double *)ctx->d_scratch;
    double *dE = p; p += K * N;
    double *de0 = p; p += N;
    double *dav = p; p += N;
    double *dbv = p; p += N;
    double *dkv = p; p += NN;
    double *dnv = p; p += NN;
    double *dgv = p; p += NN * B;
    double *dR = p; p += T * N;
    double *dcolp = p; p += (size_t)row_blocks * N;
    double *dslab = p;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(dav, alpha_v, 8 * N, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dbv, beta_v, 8 * N, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dkv, kappa_v, 8 * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dnv, nu_v, 8 * NN, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dgv, gamma_v, 8 * NN * B, hipMemcpyHostToDevice, st));
    for (int step = 0; step < n_steps; ++step) {      // variational parameters stay on the device between steps
        hipLaunchKernelGGL(k_vb_factors, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, dav, dbv, dkv, dnv, dgv, dE, de0);
        NHP_HIP(ctx, hipGetLastError());
        // GEMM-1: Z = e0 ⊕ G·E, R = data / Z, column sums of R
        gemm_args g1{};
        g1.A = ds->d_conv; g1.lda = T; g1.B = dE; g1.ldb = K; g1.M = (int)T; g1.N = (int)N; g1.K = (int)K; g1.k_chunk = (int)K;
        g1.base = de0; g1.dataT = ds->d_dataT; g1.out = dR; g1.partials = dcolp;
        launch_gemm<true, EPI_VB_Z>(g1, 1, st, bm);
        NHP_HIP(ctx, hipGetLastError());
        // GEMM-2: slabs_z = Gᵀ·R over T-chunk z
        gemm_args g2{};
        g2.A = ds->d_conv; g2.lda = T; g2.B = dR; g2.ldb = T; g2.M = (int)K; g2.N = (int)N; g2.K = (int)T; g2.k_chunk = k_chunk;
        g2.out = dslab;
        launch_gemm<false, EPI_SLAB>(g2, splits, st);
        NHP_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_vb_baseline, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, (int)N, row_blocks, dcolp, de0,
                           alpha0, beta0, (double)T * dt, dav, dbv);
        hipLaunchKernelGGL(k_vb_finish, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, splits, dslab, dE,
                           ds->d_colsum, kappa, nu, gamma, dkv, dnv, dgv);
        NHP_HIP(ctx, hipGetLastError());
    }
    NHP_HIP(ctx, hipMemcpyAsync(alpha_v, dav, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(beta_v, dbv, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(kappa_v, dkv, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(nu_v, dnv, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(gamma_v, dgv, 8 * NN * B, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

// DiscreteLogGaussianCoxProcess(x, λ, Σ, m, dt) as the baseline of this dataset's process (src/baselines.jl:461-509):
// builds the per-bin baseline intensity on the device; later calls that pass lambda0 = NULL use it.
extern "C" nhp_status nhp_disc_set_lgcp_baseline(nhp_ctx *ctx, nhp_disc_dataset *ds, const double *grid_x, int32_t grid_n,
                                                 const double *lam, double dt)
{
    if (!ctx || !ds || !grid_x || !lam || grid_n < 2) return NHP_EINVAL;
    for (int g = 0; g + 1 < grid_n; ++g)
        if (!(grid_x[g + 1] > grid_x[g])) { nhp_set_error(ctx, "grid points must be strictly increasing"); return NHP_EDOMAIN; }
    // the interpolator throws outside [x[1], x[end]] (src/utils/interpolation.jl:27); bins are evaluated at 1..T
    if (grid_x[0] > 1.0 || grid_x[grid_n - 1] < (double)ds->T) {
        nhp_set_error(ctx, "bin times 1..%lld fall outside the grid support [%g, %g]", (long long)ds->T, grid_x[0], grid_x[grid_n - 1]);
        return NHP_EDOMAIN;
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, T = (size_t)ds->T, G = (size_t)grid_n;
    if (!ds->d_baseT && hipMalloc(&ds->d_baseT, 8 * T * N) != hipSuccess) { nhp_set_error(ctx, "out of device memory (baseline matrix)"); return NHP_ENOMEM; }
    if (!ds->d_base_counts && hipMalloc(&ds->d_base_counts, 4 * T * N) != hipSuccess) { nhp_set_error(ctx, "out of device memory (baseline counts)"); return NHP_ENOMEM; }
    ds->h_grid_x.assign(grid_x, grid_x + grid_n);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (G + G * N)));
    double *dx = (double *)ctx->d_scratch, *dl = dx + G;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(dx, grid_x, 8 * G, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dl, lam, 8 * G * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_disc_base_interp, dim3((unsigned)((T + 255) / 256), (unsigned)N), dim3(256), 0, st, dx, grid_n, dl, dt, ds->T, ds->d_baseT);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

// loglikelihood(p::DiscreteLogGaussianCoxProcess, data, node, y) for all nodes (src/baselines.jl:571-584), data =
// the per-bin baseline counts parents[:, :, 1] of the latest nhp_disc_resample_parents on this dataset; cand [G*N]
// holds exp.(m .+ y) per node.
extern "C" nhp_status nhp_disc_lgcp_loglik(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *cand, double dt, double *ll)
{
    if (!ctx || !ds || !cand || !ll) return NHP_EINVAL;
    if (!ds->d_base_counts || !ds->base_counts_valid) { nhp_set_error(ctx, "lgcp_loglik: run resample_parents with the LGCP baseline first"); return NHP_EINVAL; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, G = ds->h_grid_x.size();
    // range(p) = x[1] : dt : x[end] - dt must have T points inside the support
    if (ds->h_grid_x[0] + (double)(ds->T - 1) * dt > ds->h_grid_x[G - 1]) { nhp_set_error(ctx, "lgcp_loglik: bins fall outside the grid support"); return NHP_EDOMAIN; }
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (G + G * N + N)));
    double *dx = (double *)ctx->d_scratch, *dc = dx + G, *dll = dc + G * N;
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(dx, ds->h_grid_x.data(), 8 * G, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(dc, cand, 8 * G * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_disc_lgcp_ll, dim3((unsigned)N), dim3(256), 0, st, ds->d_base_counts, ds->T, (int)G, dx, dc, dt, dll);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(ll, dll, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

