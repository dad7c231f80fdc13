// Projected L-BFGS on a box with the optimizer's state on the device (shared by nhp_cont_mle_run and nhp_disc_mle_run).
//
// Variables at a bound whose gradient points outward are held: the quasi-Newton model works on the masked gradient q, history
// pairs are stored masked (s = x_new - x, y = the free part of g_new - g), the direction is masked again, the step is projected
// back onto the box and accepted by the Armijo rule along the projected path (backtracking by halves; the first trial is the
// unit step).  The two-loop recursion runs in COEFFICIENT space ("vector-free" L-BFGS): the direction is a combination of the
// 2·HIST + 1 vectors {s_i, y_i, q}, whose coefficients follow from their dot products alone -- the products among stored pairs
// are kept on the host and only the 2·HIST + 1 products of the new q (one fused pass) and, after a step, of the new pair (one
// more) are computed; the direction is then assembled in a single pass.  Stopping rule: the reference's mle! callback,
// |f_k - f_{k-1}| < f_abstol (src/continuous.jl:168-181, src/discrete.jl:247-258), or a zero projected gradient.
//
// The objective is given as `eval(d_x, d_g)`: enqueue on ctx->stream the evaluation at the DEVICE vector d_x, leaving the
// log-likelihood in ctx->d_results[0] and g = -∇ll in d_g (f = -ll is minimised); asynchronous.
#pragma once
#include <algorithm>
#include <cmath>

#include "nhp_internal.h"

namespace {

constexpr int HIST = 8;               // limited-memory pairs
constexpr int NB = 2 * HIST;          // stored vectors: s_0..s_{HIST-1}, y_0..y_{HIST-1} (contiguous)
constexpr int NACC = 2 * NB + 3;      // u·b_j (NB), u·u, v·b_j (NB), v·v, u·v
constexpr int RBLK = 512;             // workgroups of a reduction

__device__ __forceinline__ bool held_at(double xi, double gi, double lo, double hi)
{
    return (xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0);       // g = ∇f, f minimised
}

// part[blk][k]: this block's share of u·b_j (k = j), u·u (NB), and -- with v -- v·b_j (NB + 1 + j), v·v, u·v; b_j = base + j·n
__global__ __launch_bounds__(256) void k_mle_multidot(const double *__restrict__ u, const double *__restrict__ v,
                                                      const double *__restrict__ base, int64_t n, double *__restrict__ part)
{
    __shared__ double red[NHP_WAVES][NACC];
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double ui = u[i], vi = v ? v[i] : 0.0;
        double b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = base[(size_t)j * n + i];
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc[j] += ui * b[j]; acc[NB + 1 + j] += vi * b[j]; }
        acc[NB] += ui * ui; acc[2 * NB + 1] += vi * vi; acc[2 * NB + 2] += ui * vi;
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const double w = nhp_wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double t = 0.0;
        for (int w = 0; w < NHP_WAVES; ++w) t += red[w][threadIdx.x];
        part[(size_t)blockIdx.x * NACC + threadIdx.x] = t;
    }
}

// out[k] = Σ_blk part[blk][k] (k = blockIdx.x < NACC), the RBLK partials of one product summed by one workgroup in a fixed
// order; out[NACC] = the evaluation's log-likelihood (riding along: one download).  (One 64-thread workgroup walking all
// RBLK x NACC partials took 117 us a call -- a third of an optimizer step.)
__global__ __launch_bounds__(256) void k_mle_multidot_final(const double *__restrict__ part, int nblk, const double *__restrict__ ll,
                                                           double *__restrict__ out)
{
    __shared__ double red[NHP_WAVES];
    const int k = blockIdx.x;
    double t = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) t += part[(size_t)b * NACC + k];
    t = nhp_block_sum_n<NHP_WAVES>(t, red);
    if (threadIdx.x == 0) {
        out[k] = t;
        if (k == 0) out[NACC] = ll ? *ll : 0.0;
    }
}

// q = g on the free variables, 0 on those held at a bound
__global__ __launch_bounds__(256) void k_mle_masked(double *__restrict__ q, const double *__restrict__ g, const double *__restrict__ x,
                                                    double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        q[i] = held_at(x[i], g[i], lo, hi) ? 0.0 : g[i];
}

struct mle_coef { double q, b[NB]; };

// d = (coef.q·q + Σ_j coef.b[j]·b_j) on the free variables, 0 on the held ones
__global__ __launch_bounds__(256) void k_mle_combine(double *__restrict__ d, const double *__restrict__ q, const double *__restrict__ base,
                                                     mle_coef c, const double *__restrict__ g, const double *__restrict__ x,
                                                     double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double v = c.q * q[i];
#pragma unroll
        for (int j = 0; j < NB; ++j) v += c.b[j] * base[(size_t)j * n + i];
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

// y = g_new - g on the variables that were free at (x, g), 0 on the held ones
__global__ __launch_bounds__(256) void k_mle_ydiff(double *__restrict__ y, const double *__restrict__ gn, const double *__restrict__ g,
                                                   const double *__restrict__ x, double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = held_at(x[i], g[i], lo, hi) ? 0.0 : gn[i] - g[i];
}

struct mle_state {
    nhp_ctx *ctx;
    int64_t P;
    double *d_base = nullptr;                      // the NB stored vectors
    double *d_part = nullptr, *d_scal = nullptr;   // [RBLK][NACC], [NACC + 1]
    double *h_scal = nullptr;                      // pinned [NACC + 1]
    dim3 grid;
};

// the products of u (and v) with the stored vectors -> host, together with the log-likelihood of the evaluation enqueued
// before (with_ll): ONE synchronisation
nhp_status multidot(mle_state &s, const double *u, const double *v, bool with_ll, double *out /* [NACC + 1] */)
{
    hipStream_t st = s.ctx->stream;
    hipLaunchKernelGGL(k_mle_multidot, dim3(RBLK), dim3(256), 0, st, u, v, (const double *)s.d_base, s.P, s.d_part);
    hipLaunchKernelGGL(k_mle_multidot_final, dim3(NACC), dim3(256), 0, st, (const double *)s.d_part, RBLK,
                       with_ll ? (const double *)s.ctx->d_results : (const double *)nullptr, s.d_scal);
    NHP_HIP(s.ctx, hipGetLastError());
    NHP_HIP(s.ctx, hipMemcpyAsync(s.h_scal, s.d_scal, 8 * (NACC + 1), hipMemcpyDeviceToHost, st));
    NHP_HIP(s.ctx, hipStreamSynchronize(st));
    for (int k = 0; k <= NACC; ++k) out[k] = s.h_scal[k];
    return NHP_OK;
}


// x [P] (host): the start on entry (clamped to the box), the minimiser on return; the last evaluation enqueued is at it
template <class Eval>
nhp_status nhp_lbfgs_box(nhp_ctx *ctx, int64_t P, double lower, double upper, double f_abstol, int32_t max_steps, Eval &&eval, double *x,
                         double *loss, int32_t *steps_out, int32_t *converged_out, int32_t *evals_out)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    mle_state s{ctx, P};
    s.grid = dim3((unsigned)std::min<int64_t>(2048, (P + 255) / 256));

    // x, x_new, g, g_new, q, d + HIST pairs (s_i, y_i): (6 + 2·HIST)·P doubles (370 MB at N = 1024)
    double *buf = nullptr;
    const size_t nvec = 6 + NB;
    if (hipMalloc((void **)&buf, 8 * (nvec * (size_t)P + (size_t)RBLK * NACC + NACC + 1)) != hipSuccess) {
        nhp_set_error(ctx, "out of device memory (mle! state)");
        return NHP_ENOMEM;
    }
    struct guard { double *b; double *h; ~guard() { (void)hipFree(b); if (h) (void)hipHostFree(h); } } g_{buf, nullptr};
    if (hipHostMalloc((void **)&s.h_scal, 8 * (NACC + 1)) != hipSuccess) { nhp_set_error(ctx, "out of pinned memory"); return NHP_ENOMEM; }
    g_.h = s.h_scal;
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

    NHP_TRY(eval(d_x, d_g)); ++evals;
    {
        double ll = 0.0;
        NHP_TRY(nhp_ctx_fetch(ctx, 0, 1, &ll));
        f = -ll;
    }
    if (!std::isfinite(f)) { nhp_set_error(ctx, "mle!: the objective is not finite at the starting point"); return NHP_EDOMAIN; }

    for (int it = 0; it < max_steps; ++it) {
        // ---- the masked gradient and its products with the stored vectors (one pass, one synchronisation)
        hipLaunchKernelGGL(k_mle_masked, s.grid, dim3(256), 0, st, d_q, (const double *)d_g, (const double *)d_x, lower, upper, P);
        NHP_TRY(multidot(s, d_q, nullptr, false, sc));
        const double qq = sc[NB];
        if (!(qq > 0.0)) { converged = true; break; }          // projected gradient is zero: a stationary point of the box problem
        double qs[HIST], qy[HIST];                             // q·s_j, q·y_j by slot
        for (int j = 0; j < HIST; ++j) { qs[j] = sc[j]; qy[j] = sc[HIST + j]; }
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
            nhist = 0; gd = -qq;
            NHP_HIP(ctx, hipMemsetAsync(d_S, 0, 8 * (size_t)NB * P, st));
            for (int i = 0; i < HIST; ++i) for (int j = 0; j < HIST; ++j) ss[i][j] = sy[i][j] = yy[i][j] = 0.0;
        }
        if (nhist == 0) t = std::min(1.0, 1.0 / std::sqrt(qq));
        hipLaunchKernelGGL(k_mle_combine, s.grid, dim3(256), 0, st, d_d, (const double *)d_q, (const double *)s.d_base, c,
                           (const double *)d_g, (const double *)d_x, lower, upper, P);
        // ---- backtracking along the projected path; a trial = one fused (log-likelihood, gradient) evaluation, after which
        //      the candidate step's products ride down with the objective value
        double *s_new = d_S + (size_t)head * P, *y_new = d_Y + (size_t)head * P;
        bool accepted = false;
        double fn = 0.0;
        for (int ls = 0; ls < 60; ++ls) {
            hipLaunchKernelGGL(k_mle_step, s.grid, dim3(256), 0, st, d_xn, s_new, (const double *)d_x, (const double *)d_d, t, lower, upper, P);
            NHP_HIP(ctx, hipGetLastError());
            NHP_TRY(eval(d_xn, d_gn)); ++evals;
            // u = s_new, v = q: u·v = g·(x_new - x) (s is zero on the held variables) = the Armijo slope term
            NHP_TRY(multidot(s, s_new, d_q, true, sc));
            fn = -sc[NACC];
            const double dec = sc[2 * NB + 2];
            if (std::isfinite(fn) && fn <= f + 1e-4 * dec) { accepted = true; break; }
            t *= std::isfinite(fn) ? 0.5 : 0.1;
        }
        if (!accepted) {                                       // no decrease along the path: where we are is the answer
            NHP_HIP(ctx, hipMemsetAsync(s_new, 0, 8 * (size_t)P, st));
            break;
        }
        // ---- the new pair's products with everything stored (slot `head` holds it already: its own entries come out right
        //      once y is there too, so the s-row is taken again together with the y-row)
        hipLaunchKernelGGL(k_mle_ydiff, s.grid, dim3(256), 0, st, y_new, (const double *)d_gn, (const double *)d_g, (const double *)d_x, lower, upper, P);
        NHP_TRY(multidot(s, s_new, y_new, false, sc));
        const double s_y = sc[2 * NB + 2], y_y = sc[2 * NB + 1], s_s = sc[NB];
        if (s_y > 1e-300 && y_y > 0.0 && std::isfinite(s_y) && std::isfinite(y_y) && s_y > 1e-12 * y_y) {
            for (int j = 0; j < HIST; ++j) {
                ss[head][j] = ss[j][head] = sc[j];                     // s_new·s_j
                sy[head][j] = sc[HIST + j];                            // s_new·y_j
                sy[j][head] = sc[NB + 1 + j];                          // y_new·s_j
                yy[head][j] = yy[j][head] = sc[NB + 1 + HIST + j];     // y_new·y_j
            }
            ss[head][head] = s_s; sy[head][head] = s_y; yy[head][head] = y_y;
            head = (head + 1) % HIST; nhist = std::min(nhist + 1, HIST);
        } else {
            // the pair is not kept: its slot must multiply as zeros again (and if it held the oldest pair, that one is gone)
            NHP_HIP(ctx, hipMemsetAsync(s_new, 0, 8 * (size_t)P, st));
            NHP_HIP(ctx, hipMemsetAsync(y_new, 0, 8 * (size_t)P, st));
            for (int j = 0; j < HIST; ++j) ss[head][j] = ss[j][head] = sy[head][j] = sy[j][head] = yy[head][j] = yy[j][head] = 0.0;
            if (nhist == HIST) nhist = HIST - 1;
        }
        std::swap(d_x, d_xn); std::swap(d_g, d_gn);
        f = fn; ++steps;
        if (std::fabs(f - minloss) < f_abstol) { converged = true; break; }     // the reference's callback rule
        minloss = f;
    }
    // the model holds the last TRIAL; make it the iterate
    NHP_TRY(eval(d_x, d_gn));
    {
        double ll = 0.0;
        NHP_TRY(nhp_ctx_fetch(ctx, 0, 1, &ll));
        f = -ll;
    }
    NHP_TRY(nhp_download(ctx, x, d_x, 8 * (size_t)P));
    *loss = f; *steps_out = steps; *converged_out = converged ? 1 : 0;
    if (evals_out) *evals_out = evals;
    return NHP_OK;
}

}   // namespace
