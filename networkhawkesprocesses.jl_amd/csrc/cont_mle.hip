// mle!(process, data) with the optimizer's state on the device (reference: src/continuous.jl:144-198).
//
// The reference hands Optim's Fminbox(BFGS) an objective without a gradient: 2P log-likelihood calls per finite-difference
// gradient.  The host mirror (inference.py::mle_) feeds scipy's L-BFGS-B the analytic gradient, but at the metric size every
// objective call then moves the 2.1e6 parameters up and the gradient down over PCIe (0.35 + 1.2 ms around 0.15 ms of kernels)
// and the host-side quasi-Newton update itself costs tens of milliseconds per iteration on vectors of that length.  Here the
// iterate, the gradient and the limited-memory history never leave HBM: per iteration one fused (log-likelihood, gradient)
// evaluation per line-search trial, ~2·HIST + 6 dot products (the only things the host reads: scalars) and as many axpys.
//
// Method: projected L-BFGS on the reference's box [lower, upper]^P -- variables at a bound whose gradient points outward are
// held (their direction component is zero), the two-loop recursion runs on the masked gradient, the step is projected back
// onto the box and accepted by the Armijo rule along the projected path (backtracking by halves; the first trial is the
// unit step).  Stopping rule: the reference's callback, |f_k - f_{k-1}| < f_abstol (src/continuous.jl:168-181).  Iterates
// differ from Fminbox(BFGS)'s -- the optimum of the same objective on the same box does not.
#include <algorithm>
#include <cmath>

#include "nhp_internal.h"

namespace {

constexpr int HIST = 8;       // limited-memory pairs
constexpr int RBLK = 512;     // workgroups of a reduction

__global__ __launch_bounds__(256) void k_mle_dot2(const double *__restrict__ a, const double *__restrict__ b,
                                                  const double *__restrict__ c, const double *__restrict__ d, int64_t n,
                                                  double *__restrict__ part, const double *__restrict__ mg, const double *__restrict__ mx,
                                                  double lo, double hi)
{
    // part[blk] = Σ a·b, part[RBLK + blk] = Σ c·d over this block's grid-stride share (c null: one product only); with mg / mx
    // (gradient and iterate) the sums run over the FREE variables only (see k_mle_masked)
    __shared__ double red[2 * NHP_WAVES];
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mg) {
            const double gi = mg[i], xi = mx[i];
            if ((xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0)) continue;
        }
        s0 += a[i] * b[i];
        if (c) s1 += c[i] * d[i];
    }
    nhp_block_sum2_n<NHP_WAVES>(s0, s1, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = s0; part[RBLK + blockIdx.x] = s1; }
}

__global__ __launch_bounds__(RBLK) void k_mle_dot_final(const double *__restrict__ part, double *__restrict__ out)
{
    __shared__ double red[2 * (RBLK / 64)];
    double s0 = part[threadIdx.x], s1 = part[RBLK + threadIdx.x];
    nhp_block_sum2_n<RBLK / 64>(s0, s1, red);
    if (threadIdx.x == 0) { out[0] = s0; out[1] = s1; }
}

// y += alpha * x on the free variables (held ones -- see k_mle_masked -- stay 0: the quasi-Newton model lives on the free
// subspace, built from the free components of the history pairs)
__global__ __launch_bounds__(256) void k_mle_axpy(double *__restrict__ y, const double *__restrict__ x, double alpha,
                                                  const double *__restrict__ g, const double *__restrict__ xc, double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double gi = g[i], xi = xc[i];
        const bool held = (xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0);
        if (!held) y[i] += alpha * x[i];
    }
}

// q = g on the free variables, 0 on those held at a bound (x at lower with g > 0, or at upper with g < 0: g = ∇f, f minimised)
__global__ __launch_bounds__(256) void k_mle_masked(double *__restrict__ q, const double *__restrict__ g, const double *__restrict__ x,
                                                    double lo, double hi, double scale, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double gi = g[i], xi = x[i];
        const bool held = (xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0);
        q[i] = held ? 0.0 : scale * gi;
    }
}

// d = -r on the free variables (mask as above), 0 on the held ones
__global__ __launch_bounds__(256) void k_mle_direction(double *__restrict__ d, const double *__restrict__ g, const double *__restrict__ x,
                                                       double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double gi = g[i], xi = x[i];
        const bool held = (xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0);
        d[i] = held ? 0.0 : -d[i];
    }
}

// xn = clamp(x + t d)
__global__ __launch_bounds__(256) void k_mle_step(double *__restrict__ xn, const double *__restrict__ x, const double *__restrict__ d,
                                                  double t, double lo, double hi, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i] + t * d[i];
        xn[i] = v < lo ? lo : (v > hi ? hi : v);
    }
}

// out = sign * in  (gradient of f = -ll), or a - b
__global__ __launch_bounds__(256) void k_mle_scale_copy(double *__restrict__ out, const double *__restrict__ in, double sign, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = sign * in[i];
}
__global__ __launch_bounds__(256) void k_mle_diff(double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] - b[i];
}

struct mle_state {
    nhp_ctx *ctx;
    nhp_comm *comm;
    const nhp_cont_dataset *ds;
    nhp_cont_model *m;
    int32_t flags;
    int64_t P;
    double *d_part = nullptr, *d_scal = nullptr;   // [2·RBLK], [2]
    double *h_scal = nullptr;                      // pinned [2]
    dim3 grid;
};

// (a·b, c·d) -> host (one synchronisation); mg, mx: over the free variables of (gradient mg, iterate mx) only
nhp_status dots(mle_state &s, const double *a, const double *b, const double *c, const double *d, double *ab, double *cd,
                const double *mg = nullptr, const double *mx = nullptr, double lo = 0.0, double hi = 0.0)
{
    hipStream_t st = s.ctx->stream;
    hipLaunchKernelGGL(k_mle_dot2, dim3(RBLK), dim3(256), 0, st, a, b, c, d, s.P, s.d_part, mg, mx, lo, hi);
    hipLaunchKernelGGL(k_mle_dot_final, dim3(1), dim3(RBLK), 0, st, s.d_part, s.d_scal);
    NHP_HIP(s.ctx, hipGetLastError());
    NHP_HIP(s.ctx, hipMemcpyAsync(s.h_scal, s.d_scal, 16, hipMemcpyDeviceToHost, st));
    NHP_HIP(s.ctx, hipStreamSynchronize(st));
    *ab = s.h_scal[0];
    if (cd) *cd = s.h_scal[1];
    return NHP_OK;
}

// params!(process, x) from a DEVICE vector, then f = -ll and g = -∇ll at it (g stays on the device)
nhp_status evaluate(mle_state &s, const double *d_x, double *d_g, double *f)
{
    nhp_ctx *ctx = s.ctx;
    nhp_cont_model *m = s.m;
    hipStream_t st = ctx->stream;
    const size_t N = (size_t)m->N, NN = N * N;
    const size_t nb = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS ? N : N * (size_t)m->grid_n;
    const size_t nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    ++m->version;
    NHP_HIP(ctx, hipMemcpyAsync(m->d_lambda0, d_x, 8 * nb, hipMemcpyDeviceToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_p1, d_x + nb, 8 * NN, hipMemcpyDeviceToDevice, st));
    if (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL) NHP_HIP(ctx, hipMemcpyAsync(m->d_p2, d_x + nb + NN, 8 * NN, hipMemcpyDeviceToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(m->d_W, d_x + nb + nimp, 8 * NN, hipMemcpyDeviceToDevice, st));
    double *d_grad = nullptr;
    NHP_TRY(nhp_grad_enqueue_reduced(ctx, s.comm, s.ds, m, s.flags, s.P, &d_grad));
    hipLaunchKernelGGL(k_mle_scale_copy, s.grid, dim3(256), 0, st, d_g, d_grad, -1.0, s.P);
    NHP_HIP(ctx, hipGetLastError());
    double ll = 0.0;
    NHP_TRY(nhp_ctx_fetch(ctx, 0, 1, &ll));
    *f = -ll;
    return NHP_OK;
}

}   // namespace

extern "C" nhp_status nhp_cont_mle_run(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, nhp_cont_model *m, int32_t flags,
                                       double lower, double upper, double f_abstol, int32_t max_steps, double *x, int64_t P,
                                       double *loss, int32_t *steps_out, int32_t *converged_out, int32_t *evals_out)
{
    if (!ctx || !ds || !m || !x || !loss || !steps_out || !converged_out) return NHP_EINVAL;
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (m->has_A) { nhp_set_error(ctx, "mle! is defined for ContinuousStandardHawkesProcess (src/continuous.jl:144)"); return NHP_EINVAL; }
    if (!(lower < upper) || max_steps < 0) return NHP_EDOMAIN;
    {
        const size_t N = (size_t)m->N, NN = N * N;
        const size_t nb = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS ? N : N * (size_t)m->grid_n;
        const size_t nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
        if ((size_t)P != nb + nimp + NN) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    mle_state s{ctx, comm, ds, m, flags, P};
    s.grid = dim3((unsigned)std::min<int64_t>(2048, (P + 255) / 256));

    // x, x_new, g, g_new, d + HIST pairs (s_i, y_i): (5 + 2·HIST)·P doubles (353 MB at N = 1024)
    double *buf = nullptr;
    const size_t nvec = 5 + 2 * HIST;
    if (hipMalloc((void **)&buf, 8 * (nvec * (size_t)P + 2 * RBLK + 2)) != hipSuccess) { nhp_set_error(ctx, "out of device memory (mle! state)"); return NHP_ENOMEM; }
    struct guard { double *b; double *h; ~guard() { (void)hipFree(b); if (h) (void)hipHostFree(h); } } g_{buf, nullptr};
    if (hipHostMalloc((void **)&s.h_scal, 16) != hipSuccess) { nhp_set_error(ctx, "out of pinned memory"); return NHP_ENOMEM; }
    g_.h = s.h_scal;
    double *d_x = buf, *d_xn = d_x + P, *d_g = d_xn + P, *d_gn = d_g + P, *d_d = d_gn + P, *d_S = d_d + P, *d_Y = d_S + (size_t)HIST * P;
    s.d_part = d_Y + (size_t)HIST * P; s.d_scal = s.d_part + 2 * RBLK;

    NHP_HIP(ctx, hipMemcpyAsync(d_xn, x, 8 * (size_t)P, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_mle_step, s.grid, dim3(256), 0, st, d_x, d_xn, d_xn, 0.0, lower, upper, P);     // x = clamp(guess)
    NHP_HIP(ctx, hipGetLastError());

    double f = 0.0, minloss = INFINITY;
    int evals = 0, steps = 0, nhist = 0, head = 0;        // history slots [head - nhist, head) modulo HIST, newest last
    bool converged = false;
    double rho[HIST], gamma = 1.0;
    NHP_TRY(evaluate(s, d_x, d_g, &f)); ++evals;
    if (!std::isfinite(f)) { nhp_set_error(ctx, "mle!: the objective is not finite at the starting point"); return NHP_EDOMAIN; }

    for (int it = 0; it < max_steps; ++it) {
        // ---- direction: two-loop recursion on the masked gradient
        hipLaunchKernelGGL(k_mle_masked, s.grid, dim3(256), 0, st, d_d, d_g, d_x, lower, upper, 1.0, P);
        double qq = 0.0;
        NHP_TRY(dots(s, d_d, d_d, nullptr, nullptr, &qq, nullptr));
        if (!(qq > 0.0)) { converged = true; break; }          // projected gradient is zero: a stationary point of the box problem
        double alpha[HIST];
        // curvature of every kept pair on the CURRENT free subspace (the pairs were taken on other free sets): a pair without
        // positive curvature there sits this iteration out
        for (int k = 0; k < nhist; ++k) {
            const int i = ((head - 1 - k) % HIST + HIST) % HIST;
            double sy = 0.0, yy = 0.0;
            NHP_TRY(dots(s, d_S + (size_t)i * P, d_Y + (size_t)i * P, d_Y + (size_t)i * P, d_Y + (size_t)i * P, &sy, &yy, d_g, d_x, lower, upper));
            rho[i] = (sy > 1e-300 && yy > 0.0 && sy > 1e-12 * yy) ? 1.0 / sy : 0.0;
            if (k == 0) gamma = rho[i] > 0.0 ? sy / yy : 1.0;
        }
        for (int k = 0; k < nhist; ++k) {                      // newest -> oldest
            const int i = ((head - 1 - k) % HIST + HIST) % HIST;
            double sq = 0.0;
            NHP_TRY(dots(s, d_S + (size_t)i * P, d_d, nullptr, nullptr, &sq, nullptr));
            alpha[i] = rho[i] * sq;
            hipLaunchKernelGGL(k_mle_axpy, s.grid, dim3(256), 0, st, d_d, d_Y + (size_t)i * P, -alpha[i], d_g, d_x, lower, upper, P);
        }
        if (nhist > 0) hipLaunchKernelGGL(k_mle_scale_copy, s.grid, dim3(256), 0, st, d_d, d_d, gamma, P);
        for (int k = nhist - 1; k >= 0; --k) {                 // oldest -> newest
            const int i = ((head - 1 - k) % HIST + HIST) % HIST;
            double yr = 0.0;
            NHP_TRY(dots(s, d_Y + (size_t)i * P, d_d, nullptr, nullptr, &yr, nullptr));
            hipLaunchKernelGGL(k_mle_axpy, s.grid, dim3(256), 0, st, d_d, d_S + (size_t)i * P, alpha[i] - rho[i] * yr, d_g, d_x, lower, upper, P);
        }
        hipLaunchKernelGGL(k_mle_direction, s.grid, dim3(256), 0, st, d_d, d_g, d_x, lower, upper, P);
        double gd = 0.0;
        NHP_TRY(dots(s, d_g, d_d, nullptr, nullptr, &gd, nullptr));
        double t = 1.0;
        if (!(gd < 0.0) || !std::isfinite(gd)) {               // not a descent direction: steepest descent, history dropped
            hipLaunchKernelGGL(k_mle_masked, s.grid, dim3(256), 0, st, d_d, d_g, d_x, lower, upper, -1.0, P);
            nhist = 0; gd = -qq;
        }
        if (nhist == 0) t = std::min(1.0, 1.0 / std::sqrt(qq));
        // ---- backtracking along the projected path
        bool accepted = false;
        double fn = 0.0;
        for (int ls = 0; ls < 60; ++ls) {
            hipLaunchKernelGGL(k_mle_step, s.grid, dim3(256), 0, st, d_xn, d_x, d_d, t, lower, upper, P);
            NHP_HIP(ctx, hipGetLastError());
            NHP_TRY(evaluate(s, d_xn, d_gn, &fn)); ++evals;
            hipLaunchKernelGGL(k_mle_diff, s.grid, dim3(256), 0, st, d_S + (size_t)head * P, d_xn, d_x, P);     // s = x_new - x (kept if accepted)
            double dec = 0.0;
            NHP_TRY(dots(s, d_g, d_S + (size_t)head * P, nullptr, nullptr, &dec, nullptr));
            if (std::isfinite(fn) && fn <= f + 1e-4 * dec) { accepted = true; break; }
            t *= std::isfinite(fn) ? 0.5 : 0.1;
        }
        if (!accepted) break;                                  // no decrease along the path: where we are is the answer
        // ---- history pair (the slot `head` already holds s)
        hipLaunchKernelGGL(k_mle_diff, s.grid, dim3(256), 0, st, d_Y + (size_t)head * P, d_gn, d_g, P);
        double sy = 0.0, yy = 0.0;
        NHP_TRY(dots(s, d_S + (size_t)head * P, d_Y + (size_t)head * P, d_Y + (size_t)head * P, d_Y + (size_t)head * P, &sy, &yy));
        if (sy > 1e-300 && yy > 0.0 && std::isfinite(sy) && std::isfinite(yy) && sy > 1e-12 * yy) {
            rho[head] = 1.0 / sy; gamma = sy / yy;
            head = (head + 1) % HIST; nhist = std::min(nhist + 1, HIST);
        } else if (nhist == HIST) {
            nhist = HIST - 1;                                  // the slot just overwritten was the oldest pair: it is gone
        }
        std::swap(d_x, d_xn); std::swap(d_g, d_gn);
        f = fn; ++steps;
        if (std::fabs(f - minloss) < f_abstol) { converged = true; break; }     // the reference's callback rule
        minloss = f;
    }
    // the model holds the last TRIAL; make it the iterate
    {
        double dummy = 0.0;
        NHP_TRY(evaluate(s, d_x, d_gn, &dummy));
        f = dummy;
    }
    NHP_TRY(nhp_download(ctx, x, d_x, 8 * (size_t)P));
    *loss = f; *steps_out = steps; *converged_out = converged ? 1 : 0;
    if (evals_out) *evals_out = evals;
    return NHP_OK;
}
