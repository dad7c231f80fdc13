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

// ---- table-driven exp for the log-likelihood kernels ------------------------------------------------------------------
// exp(x) = 2^n · 2^(j/64) · e^r with k = rint(x·64/ln2) = 64 n + j and |r| <= ln2/128: a 64-entry table of 2^(j/64) (correctly
// rounded, kept in LDS: 512 bytes) and a degree-5 polynomial for e^r - 1 (truncation r^6/720 < 4e-17) replace the degree-13
// polynomial of nhp_exp: 12 fp64 instructions + 3 integer + one 8-byte LDS read instead of 19 fp64.  Within 2 ulp of exp
// (checked against 200-bit arithmetic for the whole range); NOT bit-identical to nhp_exp, so only the log-likelihood
// kernels use it -- the parent sampler keeps the det-math sequence it shares with the CPU checker.
static __device__ const double nhp_exp2_64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

// every thread of the workgroup calls this before a __syncthreads(); `tab` = 64 doubles of LDS
__device__ __forceinline__ void nhp_exp_tab_init(double *tab)
{
    if (threadIdx.x < 64) tab[threadIdx.x] = nhp_exp2_64[threadIdx.x];
}

__device__ __forceinline__ double nhp_exp_neg_tab(double x, const double *tab)
{
#pragma clang fp contract(off)
    const double K64 = 92.33248261689366;                 // 64 / ln 2
    const double L64_HI = 0x1.62e42ff000000p-7;           // ln 2 / 64, 32 significant bits: k·L64_HI is exact for |k| < 2^21
    const double L64_LO = -0x1.718432a1b0e26p-41;
    const double kf = __builtin_rint(x * K64);
    double r = __builtin_fma(-kf, L64_HI, x);
    r = __builtin_fma(-kf, L64_LO, r);
    const int k = (int)kf;
    const double t = tab[k & 63];
    double p;
    const double c5 = 8.3333333333333332e-03;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"                    // ((((r/120 + 1/24) r + 1/6) r + 1/2) r + 1) r = e^r - 1
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, 0.5\n\t"
        "v_fma_f64 %0, %0, %2, 1.0\n\t"
        "v_mul_f64 %0, %0, %2"
        : "=&v"(p)
        : "v"(c5), "v"(r), "s"(4.1666666666666664e-02), "s"(1.6666666666666666e-01));
    return __builtin_ldexp(__builtin_fma(t, p, t), k >> 6);
}

// The same with the argument already multiplied by 64/ln 2 (the caller folds the factor into its rate table): t = x·64/ln2,
// u = t - k in [-1/2, 1/2] and e^(u·ln2/64) - 1 as a degree-5 polynomial in u itself (coefficients (ln2/64)^i / i!: no
// multiplication by ln2/64 either); the reduced argument loses |t|·2^-53·ln2/64 absolute (3e-15 at x = -30, 8e-14 at x = -700,
// where the term is 1e-304): used by the batch kernel and the recursion, held to the oracle at 1e-11 like the rest.
__device__ __forceinline__ double nhp_exp_neg_tab_scaled(double t, const double *tab)
{
#pragma clang fp contract(off)
    const double kf = __builtin_rint(t);
    const double u = t - kf;
    const int k = (int)kf;
    const double tb = tab[k & 63];
    double p;
    const double c5 = 0x1.5d87fe78a6731p-40;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_mul_f64 %0, %0, %2"
        : "=&v"(p)
        : "v"(c5), "v"(u), "s"(0x1.3b2ab6fba4e77p-31), "s"(0x1.c6b08d704a0c0p-23), "s"(0x1.ebfbdff82c58fp-15), "s"(0x1.62e42fefa39efp-7));
    return __builtin_ldexp(__builtin_fma(tb, p, tb), k >> 6);
}

__device__ __forceinline__ double nhp_pdf_exponential_tab(double r, double dt, const double *tab)
{
#pragma clang fp contract(off)
    return r * nhp_exp_neg_tab(-(r * dt), tab);
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

// The same value from its data-only half -- lq = {logit(x), 1/(x(1-x))}, what nhp_logitnormal_data makes of a delay once per
// dataset -- and the parameters: no logarithm and no division left per evaluation (bit-identical to nhp_pdf_logitnormal).
__device__ __forceinline__ double2 nhp_logitnormal_data(double inv_dtmax, double dt)
{
#pragma clang fp contract(off)
    const double x = dt * inv_dtmax;
    if (!(x > 0.0 && x < 1.0)) return make_double2(0.0, 0.0);
    const double o = 1.0 - x;
    const double q = 1.0 / (x * o);
    return make_double2(nhp_log((x * x) * q), q);
}
__device__ __forceinline__ double nhp_pdf_logitnormal_cached(double mu, double st, double2 lq)
{
#pragma clang fp contract(off)
    const double z = (lq.x - mu) * st;
    const double e = nhp_exp_neg(-0.5 * (z * z));
    return (e * (NHP_INVSQRT2PI * st)) * lq.y;                     // (q = 0 marks a delay outside (0, Δtmax): 0)
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
