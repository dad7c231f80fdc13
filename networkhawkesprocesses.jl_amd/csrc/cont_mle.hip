// mle!(process, data) with the optimizer's state on the device (reference: src/continuous.jl:144-198).
//
// The reference hands Optim's Fminbox(BFGS) an objective without a gradient: 2P log-likelihood calls per finite-difference
// gradient.  The host mirror (inference.py::mle_) feeds scipy's L-BFGS-B the analytic gradient, but at the metric size every
// objective call then moves the 2.1e6 parameters up and the gradient down over PCIe (0.35 + 1.2 ms around 0.15 ms of kernels)
// and the host-side quasi-Newton update itself costs most of a second per iteration on vectors of that length.  Here the
// iterate, the gradient and the limited-memory history never leave HBM, and the host reads three small sets of scalars per
// iteration.
//
// Method: projected L-BFGS on the reference's box [lower, upper]^P, two-loop recursion in coefficient space (nhp_lbfgs.h).
// Iterates differ from Fminbox(BFGS)'s; what is reached is a local maximum of the same objective on the same box
// (tests/test_cont_inference_gpu.py checks it against the host optimizer).
#include "nhp_lbfgs.h"

extern "C" nhp_status nhp_cont_mle_run(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, nhp_cont_model *m, int32_t flags,
                                       double lower, double upper, double f_abstol, int32_t max_steps, double *x, int64_t P,
                                       double *loss, int32_t *steps_out, int32_t *converged_out, int32_t *evals_out)
{
    if (!ctx || !ds || !m || !x || !loss || !steps_out || !converged_out) return NHP_EINVAL;
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (m->has_A) { nhp_set_error(ctx, "mle! is defined for ContinuousStandardHawkesProcess (src/continuous.jl:144)"); return NHP_EINVAL; }
    if (!(lower < upper) || max_steps < 0) return NHP_EDOMAIN;
    const size_t N = (size_t)m->N, NN = N * N;
    const size_t nb = m->baseline_kind == NHP_BASELINE_HOMOGENEOUS ? N : N * (size_t)m->grid_n;
    const size_t nimp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    if ((size_t)P != nb + nimp + NN) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    // params!(process, x) from a DEVICE vector, then the log-likelihood (-> ctx->d_results[0]) and g = -∇ll at it.  A trial is
    // evaluated straight from the optimizer's vector -- a view of the model whose tables point into d_x ([λ0; θ | μ; τ; W] is
    // params! order: the blocks ARE the tables) -- so no parameter is copied per evaluation; the run's last call (commit)
    // puts the iterate into the model's own tables.
    auto eval = [&](const double *d_x, double *d_g, bool commit) -> nhp_status {
        hipStream_t st = ctx->stream;
        ++m->version;
        nhp_cont_model view = *m;
        if (commit) {
            NHP_HIP(ctx, hipMemcpyAsync(m->d_lambda0, d_x, 8 * nb, hipMemcpyDeviceToDevice, st));
            NHP_HIP(ctx, hipMemcpyAsync(m->d_p1, d_x + nb, 8 * NN, hipMemcpyDeviceToDevice, st));
            if (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL) NHP_HIP(ctx, hipMemcpyAsync(m->d_p2, d_x + nb + NN, 8 * NN, hipMemcpyDeviceToDevice, st));
            NHP_HIP(ctx, hipMemcpyAsync(m->d_W, d_x + nb + nimp, 8 * NN, hipMemcpyDeviceToDevice, st));
        } else {
            double *x = const_cast<double *>(d_x);                  // (read-only through the view)
            view.d_lambda0 = x; view.d_p1 = x + nb; view.d_W = x + nb + nimp;
            if (m->impulse_kind == NHP_IMPULSE_LOGITNORMAL) view.d_p2 = x + nb + NN;
        }
        double *d_grad = nullptr;
        const nhp_status rc = nhp_grad_enqueue_reduced(ctx, comm, ds, &view, flags, P, &d_grad);
        m->rec_version = view.rec_version; m->rec_ds = view.rec_ds; m->rec_cut = view.rec_cut;    // (the recursive route's cached bound)
        NHP_TRY(rc);
        hipLaunchKernelGGL(k_mle_neg, dim3((unsigned)std::min<int64_t>(2048, (P + 255) / 256)), dim3(256), 0, st, d_g, (const double *)d_grad, P);
        NHP_HIP(ctx, hipGetLastError());
        return NHP_OK;
    };
    return nhp_lbfgs_box(ctx, P, lower, upper, f_abstol, max_steps, eval, x, loss, steps_out, converged_out, evals_out);
}

// ---- diagnostics: the optimizer alone on a separable quadratic (tools/dbg/lbfgsquad.py) ------------------------------------
// f(x) = ½ Σ h_i (x_i - c_i)² on the box: the iteration counts of nhp_lbfgs_box can be held against scipy's L-BFGS-B without
// a likelihood in between.
__global__ __launch_bounds__(256) void k_probe_quad(const double *__restrict__ x, const double *__restrict__ h, const double *__restrict__ c,
                                                    int64_t n, double *__restrict__ g, double *__restrict__ part)
{
    __shared__ double red[NHP_WAVES];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double d = x[i] - c[i];
        g[i] = h[i] * d;
        acc += 0.5 * h[i] * d * d;
    }
    acc = nhp_block_sum_n<NHP_WAVES>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_probe_quad_final(const double *__restrict__ part, int nblk, double *__restrict__ out)
{
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < nblk; ++b) t += part[b];
        *out = -t;                                                  // the framework maximises "ll" = -f
    }
}

extern "C" nhp_status nhp_probe_lbfgs(nhp_ctx *ctx, int64_t n, const double *h, const double *c, double lower, double upper, double f_abstol,
                                      int32_t max_steps, double *x, double *loss, int32_t *steps_out, int32_t *converged_out, int32_t *evals_out)
{
    if (!ctx || n < 1 || !h || !c || !x || !loss || !steps_out || !converged_out) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    double *d_h = nullptr;
    const int nblk = 64;
    if (hipMalloc((void **)&d_h, 8 * (2 * (size_t)n + nblk)) != hipSuccess) return NHP_ENOMEM;
    double *d_c = d_h + n, *d_part = d_c + n;
    NHP_HIP(ctx, hipMemcpyAsync(d_h, h, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    NHP_HIP(ctx, hipMemcpyAsync(d_c, c, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    auto eval = [&](const double *d_x, double *d_g, bool) -> nhp_status {
        hipLaunchKernelGGL(k_probe_quad, dim3(nblk), dim3(256), 0, ctx->stream, d_x, (const double *)d_h, (const double *)d_c, n, d_g, d_part);
        hipLaunchKernelGGL(k_probe_quad_final, dim3(1), dim3(64), 0, ctx->stream, (const double *)d_part, nblk, ctx->d_results);
        NHP_HIP(ctx, hipGetLastError());
        return NHP_OK;
    };
    const nhp_status rc = nhp_lbfgs_box(ctx, n, lower, upper, f_abstol, max_steps, eval, x, loss, steps_out, converged_out, evals_out);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_h);
    return rc;
}
