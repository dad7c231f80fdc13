// Device-side fp64 math for the continuous kernels (gfx950).
//
// exp / log are evaluated with a FIXED sequence of IEEE-754 operations (explicit fma where
// written, nothing contracted elsewhere) so that the categorical weights of the parent
// sampler -- and therefore the sampled parent indices -- are reproducible bit for bit
// against a CPU evaluation of the same sequence (BASELINE.json: "bit-exact for
// parent-index sampling given a fixed RNG stream").  The sequence is documented in
// DESIGN.md ("det-math contract"); tests/ compare it bitwise with the checker's copy.
//
// gfx950 has no fp64 transcendental unit: exp is 1 mul + rndne + 2 fma (Cody-Waite) +
// 13 fma (Horner, degree-13 Taylor on |r| <= ln2/2) + cvt + ldexp, all full-rate DP ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NHP_INVSQRT2PI 0.3989422804014327

// One Horner step p*r + c with the coefficient in a scalar register.  Written as a three-address
// v_fma_f64 by hand: left to itself hipcc selects the two-address v_fmac_f64, whose addend is
// overwritten, and so copies every (loop-invariant) coefficient with a v_mov_b64 first -- ten extra
// VALU instructions per exp.  Same IEEE fma, so the det-math contract is untouched.
__device__ __forceinline__ double nhp_horner(double p, double r, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
    return d;
}

// The whole degree-13 Horner chain of nhp_exp as ONE asm statement.  Same thirteen IEEE fmas in the same order as thirteen
// nhp_horner() calls (det-math contract untouched), but gfx950's hazard recognizer pads every inline-asm result that the next
// instruction reads with an `s_nop` (it must assume a dst_sel / cvt-scale forwarding hazard it cannot see into): one statement
// pays that once instead of thirteen times per exponential.
__device__ __forceinline__ double nhp_horner13(double r)
{
    double p;
    const double c13 = 1.6059043836821613e-10;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %0, %0, %2, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\t"
        "v_fma_f64 %0, %0, %2, %9\n\t"
        "v_fma_f64 %0, %0, %2, %10\n\t"
        "v_fma_f64 %0, %0, %2, %11\n\t"
        "v_fma_f64 %0, %0, %2, %12\n\t"
        "v_fma_f64 %0, %0, %2, %13\n\t"
        "v_fma_f64 %0, %0, %2, %14\n\t"
        "v_fma_f64 %0, %0, %2, %14"
        : "=&v"(p)
        : "v"(c13), "v"(r), "s"(2.08767569878681e-09), "s"(2.505210838544172e-08), "s"(2.755731922398589e-07),
          "s"(2.7557319223985893e-06), "s"(2.48015873015873e-05), "s"(1.984126984126984e-04), "s"(1.388888888888889e-03),
          "s"(8.333333333333333e-03), "s"(4.1666666666666664e-02), "s"(1.6666666666666666e-01), "s"(0.5), "s"(1.0));
    return p;
}

__device__ __forceinline__ double nhp_exp(double x)
{
#pragma clang fp contract(off)
    if (!(x >= -708.0)) return (x != x) ? x : 0.0;
    if (x > 709.0) return __builtin_inf();
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = __builtin_rint(x * LOG2E);
    double r = __builtin_fma(-n, LN2_HI, x);
    r = __builtin_fma(-n, LN2_LO, r);
    double p = nhp_horner13(r);
    return __builtin_ldexp(p, (int)n);
}

// exp(x) for x <= 0 without the overflow / NaN branches (callers guarantee the range).
__device__ __forceinline__ double nhp_exp_neg(double x)
{
#pragma clang fp contract(off)
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = __builtin_rint(x * LOG2E);
    double r = __builtin_fma(-n, LN2_HI, x);
    r = __builtin_fma(-n, LN2_LO, r);
    double p = nhp_horner13(r);
    double v = __builtin_ldexp(p, (int)n);
    return (x >= -708.0) ? v : 0.0;
}

// exp(x) for x <= 0 in the log-likelihood kernels: no flush -- v_ldexp_f64 underflows by itself (gradually below -708,
// to zero below -745), so the compare and the two selects of nhp_exp_neg are saved.  Differs from nhp_exp_neg only on
// (-745, -708), by less than 1e-307: nothing a log-likelihood registers; the sampler (bit-exact contract) keeps nhp_exp_neg.
__device__ __forceinline__ double nhp_exp_neg_ll(double x)
{
#pragma clang fp contract(off)
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = __builtin_rint(x * LOG2E);
    double r = __builtin_fma(-n, LN2_HI, x);
    r = __builtin_fma(-n, LN2_LO, r);
    return __builtin_ldexp(nhp_horner13(r), (int)n);
}

__device__ __forceinline__ double nhp_log(double x)
{
#pragma clang fp contract(off)
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -__builtin_inf();
        if (hx >> 31) return __builtin_nan("");
        k -= 54;
        x *= 18014398509481984.0;
        u = (uint64_t)__double_as_longlong(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffu);
    x = __longlong_as_double((long long)u);
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// Philox4x32-10, counter (event, step), key seed -> 53-bit uniform in [0,1).
__host__ __device__ __forceinline__ double nhp_philox_uniform(uint64_t seed, uint64_t step, uint64_t event)
{
    uint32_t c0 = (uint32_t)event, c1 = (uint32_t)(event >> 32);
    uint32_t c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    uint64_t bits = ((uint64_t)c0 << 32) | c1;
    return (double)(bits >> 11) * 1.1102230246251565e-16;
}

// ---- pair evaluators: impulse pdf for one (parent, child) pair, without the weight -----

// Exponential, r = rate.  Reference: pdf(Exponential(1/θ), Δt) = r*exp(-r*Δt)
// (src/impulses.jl:106-108).  dt >= 0 is guaranteed by the sorted event order.
__device__ __forceinline__ double nhp_pdf_exponential(double r, double dt)
{
#pragma clang fp contract(off)
    return r * nhp_exp_neg(-(r * dt));
}

__device__ __forceinline__ double nhp_pdf_exponential_ll(double r, double dt)
{
#pragma clang fp contract(off)
#ifdef NHP_EXP_STUB        // timing experiment only (tools/): the pair term without its exponential
    return r * (1.0 - 1e-3 * (r * dt));
#else
    return r * nhp_exp_neg_ll(-(r * dt));
#endif
}

// Logit-normal at x = Δt/Δtmax, NOT divided by Δtmax (src/impulses.jl:174-178).
// st = sqrt(τ).  One division for q = 1/(x(1-x)), logit(x) = log(x*x*q).
__device__ __forceinline__ double nhp_pdf_logitnormal(double mu, double st, double inv_dtmax, double dt)
{
#pragma clang fp contract(off)
    double x = dt * inv_dtmax;
    if (!(x > 0.0 && x < 1.0)) return 0.0;
    double o = 1.0 - x;
    double q = 1.0 / (x * o);
    double lx = nhp_log((x * x) * q);
    double z = (lx - mu) * st;
    double e = nhp_exp_neg(-0.5 * (z * z));
    return (e * (NHP_INVSQRT2PI * st)) * q;
}
