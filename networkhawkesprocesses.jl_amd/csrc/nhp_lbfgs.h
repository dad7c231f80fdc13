// Projected L-BFGS on a box with the optimizer's state on the device (shared by nhp_cont_mle_run and nhp_disc_mle_run).
//
// Variables at a bound whose gradient points outward are held: the quasi-Newton model works on the masked gradient q, history
// pairs are stored masked (s = x_new - x, y = the free part of g_new - g), the direction is masked again, the step is projected
// back onto the box and accepted by the Armijo rule along the projected path (backtracking by halves; the first trial is the
// unit step).  The two-loop recursion runs in COEFFICIENT space ("vector-free" L-BFGS): the direction is a combination of the
// 2·HIST + 1 vectors {s_i, y_i, q}, whose coefficients follow from their dot products alone -- the products among stored pairs
// are kept on the host; after an accepted step ONE pass over the stored vectors takes the products of the new pair AND of the
// next masked gradient with them (k_mle_multidot3; slots never written since the last reset are not read), and the direction
// is then assembled in a single pass.  A line-search trial reads back one scalar pair: the objective and g·(x_new - x).
// Per step: two passes over the history, one evaluation, two synchronisations.  Stopping rule: the reference's mle! callback,
// |f_k - f_{k-1}| < f_abstol (src/continuous.jl:168-181, src/discrete.jl:247-258), or a zero projected gradient.
//
// The objective is given as `eval(d_x, d_g, commit)`: enqueue on ctx->stream the evaluation at the DEVICE vector d_x, leaving
// the log-likelihood in ctx->d_results[0] and g = -∇ll in d_g (f = -ll is minimised); asynchronous.  `commit` marks the
// run's last call -- the iterate the caller's model has to hold afterwards (a trial may be evaluated straight from d_x).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>

#include "nhp_internal.h"

namespace {

#ifndef NHP_LBFGS_HIST
#define NHP_LBFGS_HIST 8       // (variant builds: tools/dbg/mlesparse.py measured 4 / 8 / 16 pairs alike on a sparse fit)
#endif
constexpr int HIST = NHP_LBFGS_HIST;  // limited-memory pairs
constexpr int NB = 2 * HIST;          // stored vectors: s_0..s_{HIST-1}, y_0..y_{HIST-1} (contiguous)
constexpr int NACC = 3 * NB + 6;      // u·b_j, v·b_j, w·b_j (NB each), then u·u, v·v, w·w, u·v, u·w, v·w
constexpr int RBLK = 512;             // workgroups of a reduction

__device__ __forceinline__ bool held_at(double xi, double gi, double lo, double hi)
{
    return (xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0);       // g = ∇f, f minimised
}

// part[blk][k]: this block's share of the products of u, v, w (null = a zero vector) with the stored vectors b_j = base + j·n
// of the slots j < na (both halves: s_j and y_j; the other slots hold zeros and are not read) and with each other
__global__ __launch_bounds__(256) void k_mle_multidot3(const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                                                       const double *__restrict__ base, int na, int64_t n, double *__restrict__ part)
{
    __shared__ double red[NHP_WAVES][NACC];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    auto one = [&](const double ui, const double vi, const double wi, const double (&b)[NB]) {
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc[j] += ui * b[j]; acc[NB + j] += vi * b[j]; acc[2 * NB + j] += wi * b[j]; }
        acc[3 * NB] += ui * ui; acc[3 * NB + 1] += vi * vi; acc[3 * NB + 2] += wi * wi;
        acc[3 * NB + 3] += ui * vi; acc[3 * NB + 4] += ui * wi; acc[3 * NB + 5] += vi * wi;
    };
    // two consecutive elements per thread and trip: 16-byte loads of every vector (vector starts are 16-byte aligned when n is even:
    // the state is carved in units of n doubles from a hipMalloc'ed block); an odd n takes the scalar loop
    if ((n & 1) == 0) {
        const int64_t n2 = n >> 1;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
            const double2 z = make_double2(0.0, 0.0);
            const double2 ui = u ? reinterpret_cast<const double2 *>(u)[i] : z, vi = v ? reinterpret_cast<const double2 *>(v)[i] : z;
            const double2 wi = reinterpret_cast<const double2 *>(w)[i];
            double b0[NB], b1[NB];
#pragma unroll
            for (int j = 0; j < HIST; ++j) {
                const double2 s2 = j < na ? reinterpret_cast<const double2 *>(base + (size_t)j * n)[i] : z;
                const double2 y2 = j < na ? reinterpret_cast<const double2 *>(base + (size_t)(HIST + j) * n)[i] : z;
                b0[j] = s2.x; b1[j] = s2.y; b0[HIST + j] = y2.x; b1[HIST + j] = y2.y;
            }
            one(ui.x, vi.x, wi.x, b0);
            one(ui.y, vi.y, wi.y, b1);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const double ui = u ? u[i] : 0.0, vi = v ? v[i] : 0.0, wi = w[i];
            double b[NB];
#pragma unroll
            for (int j = 0; j < HIST; ++j) {
                b[j] = j < na ? base[(size_t)j * n + i] : 0.0;
                b[HIST + j] = j < na ? base[(size_t)(HIST + j) * n + i] : 0.0;
            }
            one(ui, vi, wi, b);
        }
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const double ws = nhp_wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = ws;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double t = 0.0;
        for (int q = 0; q < NHP_WAVES; ++q) t += red[q][threadIdx.x];
        part[(size_t)blockIdx.x * NACC + threadIdx.x] = t;
    }
}

// out[k] = Σ_blk part[blk][k] (k = blockIdx.x < nacc), the partials of one product summed by one workgroup in a fixed order;
// out[nacc] = the evaluation's log-likelihood (riding along: one download).  (One 64-thread workgroup walking all RBLK x NACC
// partials took 117 us a call -- a third of an optimizer step.)
__global__ __launch_bounds__(256) void k_mle_multidot_final(const double *__restrict__ part, int nblk, int nacc, const double *__restrict__ ll,
                                                           double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    const int k = blockIdx.x;
    double t = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) t += part[(size_t)b * nacc + k];
    t = nhp_block_sum_n<NHP_WAVES>(t, red);
    if (threadIdx.x == 0) {
        out[k] = t;
        if (k == 0) out[nacc] = ll ? *ll : 0.0;
    }
}

// part[blk] = this block's share of u·v (the Armijo slope term of a trial: two vectors, not the whole history)
__global__ __launch_bounds__(256) void k_mle_dot(const double *__restrict__ u, const double *__restrict__ v, int64_t n, double *__restrict__ part)
{
    __shared__ double red[NHP_WAVES];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += u[i] * v[i];
    acc = nhp_block_sum_n<NHP_WAVES>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// q = g on the free variables, 0 on those held at a bound
__global__ __launch_bounds__(256) void k_mle_masked(double *__restrict__ q, const double *__restrict__ g, const double *__restrict__ x,
                                                    double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        q[i] = held_at(x[i], g[i], lo, hi) ? 0.0 : g[i];
}

struct mle_coef { double q, b[NB]; };

// d = (coef.q·q + Σ_j coef.b[j]·b_j) on the free variables, 0 on the held ones; only the first `na` slots are read
__global__ __launch_bounds__(256) void k_mle_combine(double *__restrict__ d, const double *__restrict__ q, const double *__restrict__ base,
                                                     mle_coef c, int na, const double *__restrict__ g, const double *__restrict__ x,
                                                     double lo, double hi, int64_t n)
{
    if ((n & 1) == 0) {                                             // 16-byte loads (see k_mle_multidot3)
        const int64_t n2 = n >> 1;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
            const double2 qi = reinterpret_cast<const double2 *>(q)[i];
            double v0 = c.q * qi.x, v1 = c.q * qi.y;
#pragma unroll
            for (int j = 0; j < HIST; ++j)
                if (j < na) {
                    const double2 s2 = reinterpret_cast<const double2 *>(base + (size_t)j * n)[i];
                    const double2 y2 = reinterpret_cast<const double2 *>(base + (size_t)(HIST + j) * n)[i];
                    v0 += c.b[j] * s2.x + c.b[HIST + j] * y2.x;
                    v1 += c.b[j] * s2.y + c.b[HIST + j] * y2.y;
                }
            const double2 gi = reinterpret_cast<const double2 *>(g)[i], xi = reinterpret_cast<const double2 *>(x)[i];
            reinterpret_cast<double2 *>(d)[i] = make_double2(held_at(xi.x, gi.x, lo, hi) ? 0.0 : v0, held_at(xi.y, gi.y, lo, hi) ? 0.0 : v1);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double v = c.q * q[i];
#pragma unroll
        for (int j = 0; j < HIST; ++j)
            if (j < na) v += c.b[j] * base[(size_t)j * n + i] + c.b[HIST + j] * base[(size_t)(HIST + j) * n + i];   // (the other slots hold zeros)
        d[i] = held_at(x[i], g[i], lo, hi) ? 0.0 : v;
    }
}

// xn = clamp(x + t d), s = xn - x
__global__ __launch_bounds__(256) void k_mle_step(double *__restrict__ xn, double *__restrict__ s, const double *__restrict__ x,
                                                  const double *__restrict__ d, double t, double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double xi = x[i], v = xi + t * d[i];
        const double c = v < lo ? lo : (v > hi ? hi : v);
        xn[i] = c;
        if (s) s[i] = c - xi;
    }
}

// g = -grad (gradient of f = -ll)
__global__ __launch_bounds__(256) void k_mle_neg(double *__restrict__ out, const double *__restrict__ in, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = -in[i];
}

// after an accepted step: y = g_new - g on the variables that were free at (x, g), 0 on the held ones, and the NEXT masked
// gradient q = g_new on the variables free at (x_new, g_new) -- one pass over the five vectors
__global__ __launch_bounds__(256) void k_mle_ydiff_masked(double *__restrict__ y, double *__restrict__ qn, const double *__restrict__ gn,
                                                          const double *__restrict__ g, const double *__restrict__ x,
                                                          const double *__restrict__ xn, double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double gni = gn[i], gi = g[i];
        y[i] = held_at(x[i], gi, lo, hi) ? 0.0 : gni - gi;
        qn[i] = held_at(xn[i], gni, lo, hi) ? 0.0 : gni;
    }
}

struct mle_state {
    nhp_ctx *ctx;
    int64_t P;
    double *d_base = nullptr;                      // the NB stored vectors
    double *d_part = nullptr, *d_scal = nullptr;   // [RBLK][NACC], [NACC + 1]
    double *h_scal = nullptr;                      // pinned [NACC + 1]
    dim3 grid;
};

// the products of u, v (nullable) and w with the stored vectors of the first `na` slots and with each other -> host: ONE
// synchronisation
nhp_status multidot3(mle_state &s, const double *u, const double *v, const double *w, int na, double *out /* [NACC + 1] */)
{
    hipStream_t st = s.ctx->stream;
    hipLaunchKernelGGL(k_mle_multidot3, dim3(RBLK), dim3(256), 0, st, u, v, w, (const double *)s.d_base, na, s.P, s.d_part);
    hipLaunchKernelGGL(k_mle_multidot_final, dim3(NACC), dim3(256), 0, st, (const double *)s.d_part, RBLK, NACC, (const double *)nullptr, s.d_scal);
    NHP_HIP(s.ctx, hipGetLastError());
    NHP_HIP(s.ctx, hipMemcpyAsync(s.h_scal, s.d_scal, 8 * (NACC + 1), hipMemcpyDeviceToHost, st));
    NHP_HIP(s.ctx, hipStreamSynchronize(st));
    for (int k = 0; k <= NACC; ++k) out[k] = s.h_scal[k];
    return NHP_OK;
}

// u·v and the log-likelihood of the evaluation enqueued before -> host: the one readback of a line-search trial
nhp_status dot_with_ll(mle_state &s, const double *u, const double *v, double *uv, double *ll)
{
    hipStream_t st = s.ctx->stream;
    hipLaunchKernelGGL(k_mle_dot, dim3(RBLK), dim3(256), 0, st, u, v, s.P, s.d_part);
    hipLaunchKernelGGL(k_mle_multidot_final, dim3(1), dim3(256), 0, st, (const double *)s.d_part, RBLK, 1, (const double *)s.ctx->d_results, s.d_scal);
    NHP_HIP(s.ctx, hipGetLastError());
    NHP_HIP(s.ctx, hipMemcpyAsync(s.h_scal, s.d_scal, 8 * 2, hipMemcpyDeviceToHost, st));
    NHP_HIP(s.ctx, hipStreamSynchronize(st));
    *uv = s.h_scal[0]; *ll = s.h_scal[1];
    return NHP_OK;
}


// x [P] (host): the start on entry (clamped to the box), the minimiser on return; the last evaluation enqueued is at it
template <class Eval>
nhp_status nhp_lbfgs_box(nhp_ctx *ctx, int64_t P, double lower, double upper, double f_abstol, int32_t max_steps, Eval &&eval, double *x,
                         double *loss, int32_t *steps_out, int32_t *converged_out, int32_t *evals_out)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // NHP_TIMING=1: where a run spends its wall time outside the steps (stderr)
    static const bool timing = getenv("NHP_TIMING") && atoi(getenv("NHP_TIMING")) != 0;
    static const bool trace = getenv("NHP_MLE_TRACE") && atoi(getenv("NHP_MLE_TRACE")) != 0;     // one line per accepted step (stderr)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhp mle] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    mle_state s{ctx, P};
    s.grid = dim3((unsigned)std::min<int64_t>(2048, (P + 255) / 256));

    // x, x_new, g, g_new, q, d + HIST pairs (s_i, y_i): (6 + 2·HIST)·P doubles (370 MB at N = 1024)
    const size_t nvec = 6 + NB;
    const size_t need = 8 * (nvec * (size_t)P + (size_t)RBLK * NACC + NACC + 1);
    if (ctx->mle_cap < need) {                                  // (kept by the context between runs)
        (void)hipFree(ctx->d_mle);
        ctx->d_mle = nullptr; ctx->mle_cap = 0;
        if (hipMalloc(&ctx->d_mle, need) != hipSuccess) {
            (void)hipGetLastError();
            nhp_set_error(ctx, "out of device memory (mle! state)");
            return NHP_ENOMEM;
        }
        ctx->mle_cap = need;
    }
    if (!ctx->h_mle_scal && hipHostMalloc((void **)&ctx->h_mle_scal, 8 * (NACC + 1)) != hipSuccess) {
        ctx->h_mle_scal = nullptr;
        nhp_set_error(ctx, "out of pinned memory");
        return NHP_ENOMEM;
    }
    double *buf = (double *)ctx->d_mle;
    s.h_scal = ctx->h_mle_scal;
    double *d_x = buf, *d_xn = d_x + P, *d_g = d_xn + P, *d_gn = d_g + P, *d_q = d_gn + P, *d_d = d_q + P;
    double *d_S = d_d + P, *d_Y = d_S + (size_t)HIST * P;
    s.d_base = d_S; s.d_part = d_Y + (size_t)HIST * P; s.d_scal = s.d_part + (size_t)RBLK * NACC;
    NHP_HIP(ctx, hipMemsetAsync(d_S, 0, 8 * (size_t)NB * P, st));                                       // unused slots multiply as zeros

    NHP_HIP(ctx, hipMemcpyAsync(d_xn, x, 8 * (size_t)P, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_mle_step, s.grid, dim3(256), 0, st, d_x, (double *)nullptr, (const double *)d_xn, (const double *)d_xn, 0.0, lower, upper, P);   // x = clamp(guess)
    NHP_HIP(ctx, hipGetLastError());

    // products among the stored vectors, by slot: ss[i][j] = s_i·s_j, sy[i][j] = s_i·y_j, yy[i][j] = y_i·y_j
    double ss[HIST][HIST] = {}, sy[HIST][HIST] = {}, yy[HIST][HIST] = {};
    double sc[NACC + 1];
    double f = 0.0, minloss = INFINITY;
    int evals = 0, steps = 0, nhist = 0, head = 0;        // active slots [head - nhist, head) modulo HIST, newest last
    bool converged = false;

    lap("state + start uploaded");
    NHP_TRY(eval(d_x, d_g, false)); ++evals;
    {
        double ll = 0.0;
        NHP_TRY(nhp_ctx_fetch(ctx, 0, 1, &ll));
        f = -ll;
    }
    if (!std::isfinite(f)) { nhp_set_error(ctx, "mle!: the objective is not finite at the starting point"); return NHP_EDOMAIN; }

    lap("first evaluation");
    // the masked gradient at the start and its products (no pair stored yet: nothing but q·q is read)
    int filled = 0;                                            // slots written since the last reset (the others hold zeros)
    double qq = 0.0, qs[HIST] = {}, qy[HIST] = {};             // q·q, q·s_j, q·y_j by slot
    hipLaunchKernelGGL(k_mle_masked, s.grid, dim3(256), 0, st, d_q, (const double *)d_g, (const double *)d_x, lower, upper, P);
    NHP_TRY(multidot3(s, nullptr, nullptr, d_q, 0, sc));
    qq = sc[3 * NB + 2];
    auto drop_history = [&]() -> nhp_status {
        NHP_HIP(ctx, hipMemsetAsync(d_S, 0, 8 * (size_t)NB * P, st));
        for (int i = 0; i < HIST; ++i) {
            qs[i] = qy[i] = 0.0;
            for (int j = 0; j < HIST; ++j) ss[i][j] = sy[i][j] = yy[i][j] = 0.0;
        }
        nhist = 0; head = 0; filled = 0;
        return NHP_OK;
    };
    for (int it = 0; it < max_steps; ++it) {
        if (!(qq > 0.0)) { converged = true; break; }          // projected gradient is zero: a stationary point of the box problem
        const double qq_before = qq;
        // ---- two-loop recursion on coefficients: p = cq·q + Σ cs[j]·s_j + Σ cy[j]·y_j, starting from p = -q
        mle_coef c{};
        double *cs = c.b, *cy = c.b + HIST;
        c.q = -1.0;
        double alpha[HIST] = {}, rho[HIST] = {};
        auto slot = [&](int k) { return ((head - 1 - k) % HIST + HIST) % HIST; };      // k = 0: newest
        auto dot_s = [&](int i) {                              // p·s_i
            double t = c.q * qs[i];
            for (int j = 0; j < HIST; ++j) t += cs[j] * ss[j][i] + cy[j] * sy[i][j];
            return t;
        };
        auto dot_y = [&](int i) {                              // p·y_i
            double t = c.q * qy[i];
            for (int j = 0; j < HIST; ++j) t += cs[j] * sy[j][i] + cy[j] * yy[j][i];
            return t;
        };
        for (int k = 0; k < nhist; ++k) {                      // newest -> oldest
            const int i = slot(k);
            rho[i] = 1.0 / sy[i][i];
            alpha[i] = rho[i] * dot_s(i);
            cy[i] -= alpha[i];
        }
        if (nhist > 0) {
            const int i = slot(0);
            const double gamma = sy[i][i] / yy[i][i];
            c.q *= gamma;
            for (int j = 0; j < NB; ++j) c.b[j] *= gamma;
        }
        for (int k = nhist - 1; k >= 0; --k) {                 // oldest -> newest
            const int i = slot(k);
            cs[i] += alpha[i] - rho[i] * dot_y(i);
        }
        // g·d = q·p (q is zero where d is masked): from the coefficients, no pass over the vectors
        double gd = c.q * qq;
        for (int j = 0; j < HIST; ++j) gd += cs[j] * qs[j] + cy[j] * qy[j];
        double t = 1.0;
        if (!(gd < 0.0) || !std::isfinite(gd)) {               // not a descent direction: steepest descent, history dropped
            c = mle_coef{}; c.q = -1.0;
            gd = -qq;
            NHP_TRY(drop_history());
        }
        if (nhist == 0) t = std::min(1.0, 1.0 / std::sqrt(qq));
        hipLaunchKernelGGL(k_mle_combine, s.grid, dim3(256), 0, st, d_d, (const double *)d_q, (const double *)s.d_base, c, filled,
                           (const double *)d_g, (const double *)d_x, lower, upper, P);
        // ---- backtracking along the projected path; a trial = one fused (log-likelihood, gradient) evaluation and one readback
        //      of two scalars: the objective and the Armijo slope term g·(x_new - x) = q·s_new (s is zero on the held variables).
        //      A trial is accepted only on a path that descends to first order (dec < 0: once the projection has clipped the
        //      descent components of a quasi-Newton step, what is left may point uphill; the step is then shortened like any
        //      other failed trial) -- never with f_new > f.
        double *s_new = d_S + (size_t)head * P, *y_new = d_Y + (size_t)head * P;
        bool accepted = false;
        double fn = 0.0;
        for (int ls = 0; ls < 60; ++ls) {
            hipLaunchKernelGGL(k_mle_step, s.grid, dim3(256), 0, st, d_xn, s_new, (const double *)d_x, (const double *)d_d, t, lower, upper, P);
            NHP_HIP(ctx, hipGetLastError());
            NHP_TRY(eval(d_xn, d_gn, false)); ++evals;
            double dec = 0.0, ll = 0.0;
            NHP_TRY(dot_with_ll(s, s_new, d_q, &dec, &ll));
            fn = -ll;
            if (std::isfinite(fn) && dec < 0.0 && fn <= f + 1e-4 * dec) { accepted = true; break; }
            t *= std::isfinite(fn) ? 0.5 : 0.1;                 // (a shorter step is clipped in fewer coordinates: g·d < 0 wins in the end)
        }
        if (!accepted) {
            NHP_HIP(ctx, hipMemsetAsync(s_new, 0, 8 * (size_t)P, st));     // the slot multiplies as zeros again
            if (nhist > 0) {                                   // a quasi-Newton direction failed: the projected gradient's own path next
                NHP_TRY(drop_history());
                continue;
            }
            break;                                             // no decrease along the projected gradient: where we are is the answer
        }
        // ---- the new pair and the next masked gradient, then ONE pass over the stored vectors for the products of all three
        //      (slot `head` holds the pair already, so its own entries come out with the rest)
        hipLaunchKernelGGL(k_mle_ydiff_masked, s.grid, dim3(256), 0, st, y_new, d_q, (const double *)d_gn, (const double *)d_g, (const double *)d_x,
                           (const double *)d_xn, lower, upper, P);
        filled = std::max(filled, head + 1);
        NHP_TRY(multidot3(s, s_new, y_new, d_q, filled, sc));
        const double s_s = sc[3 * NB], y_y = sc[3 * NB + 1], s_y = sc[3 * NB + 3];
        qq = sc[3 * NB + 2];
        for (int j = 0; j < HIST; ++j) { qs[j] = sc[2 * NB + j]; qy[j] = sc[2 * NB + HIST + j]; }
        if (s_y > 1e-300 && y_y > 0.0 && std::isfinite(s_y) && std::isfinite(y_y) && s_y > 1e-12 * y_y) {
            for (int j = 0; j < HIST; ++j) {
                ss[head][j] = ss[j][head] = sc[j];                     // s_new·s_j
                sy[head][j] = sc[HIST + j];                            // s_new·y_j
                sy[j][head] = sc[NB + j];                              // y_new·s_j
                yy[head][j] = yy[j][head] = sc[NB + HIST + j];         // y_new·y_j
            }
            ss[head][head] = s_s; sy[head][head] = s_y; yy[head][head] = y_y;
            head = (head + 1) % HIST; nhist = std::min(nhist + 1, HIST);
        } else {
            // the pair is not kept: its slot must multiply as zeros again (and if it held the oldest pair, that one is gone)
            NHP_HIP(ctx, hipMemsetAsync(s_new, 0, 8 * (size_t)P, st));
            NHP_HIP(ctx, hipMemsetAsync(y_new, 0, 8 * (size_t)P, st));
            for (int j = 0; j < HIST; ++j) ss[head][j] = ss[j][head] = sy[head][j] = sy[j][head] = yy[head][j] = yy[j][head] = 0.0;
            qs[head] = qy[head] = 0.0;
            if (nhist == HIST) nhist = HIST - 1;
        }
        if (trace) fprintf(stderr, "[nhp mle] step %4d f %.9f  decrease %.3e  q.q %.3e  g.d %.3e  t %.3e  pairs %d  s.y %.3e  y.y %.3e\n", steps, fn, f - fn, qq_before, gd, t, nhist, s_y, y_y);
        std::swap(d_x, d_xn); std::swap(d_g, d_gn);
        f = fn; ++steps;
        if (std::fabs(f - minloss) < f_abstol) { converged = true; break; }     // the reference's callback rule
        minloss = f;
    }
    lap("steps");
    // the model holds the last TRIAL; make it the iterate
    NHP_TRY(eval(d_x, d_gn, true));
    {
        double ll = 0.0;
        NHP_TRY(nhp_ctx_fetch(ctx, 0, 1, &ll));
        f = -ll;
    }
    NHP_TRY(nhp_download(ctx, x, d_x, 8 * (size_t)P));
    lap("last evaluation + download");
    *loss = f; *steps_out = steps; *converged_out = converged ? 1 : 0;
    if (evals_out) *evals_out = evals;
    return NHP_OK;
}

}   // namespace
