/*
 * oracle/nhp_detmath.h -- TEST INFRASTRUCTURE (checker), not product code.
 *
 * Deterministic fp64 exp / log / Philox used by the oracle's "det" mode.
 *
 * Why this exists: BASELINE.json asks for parent indices that are bit-exact for
 * a fixed RNG stream.  libm's exp() and the GPU's exp() may differ in the last
 * ulp, which would make that a probabilistic statement.  Instead the oracle and
 * the HIP kernels both evaluate exp/log with the SAME sequence of IEEE-754
 * operations (explicit fma where written, no contraction anywhere else), so the
 * categorical weights -- and therefore the sampled parent indices -- agree bit
 * for bit by construction.  The kernels carry their own restatement of this
 * sequence (networkhawkesprocesses.jl_amd/csrc/nhp_math.h); tests/ compare the
 * two bitwise on random inputs.
 *
 * Build with -ffp-contract=off (see oracle/Makefile): every '*' followed by '+'
 * below is two roundings unless written as fma().
 */
#ifndef NHP_DETMATH_H
#define NHP_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* exp(x): n = rint(x*log2e); r = x - n*ln2 (hi/lo, fma); degree-13 Taylor in
 * Horner form with fma; scale by 2^n.  Flush below -708 (result would be
 * < 3.4e-308), +inf above 709.  Max error ~1 ulp. */
static inline double nhp_det_exp(double x)
{
    if (!(x >= -708.0)) return (x != x) ? x : 0.0;
    if (x > 709.0) return INFINITY;
    const double LOG2E  = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = rint(x * LOG2E);
    double r = fma(-n, LN2_HI, x);
    r = fma(-n, LN2_LO, r);
    double p = 1.6059043836821613e-10;            /* 1/13! */
    p = fma(p, r, 2.08767569878681e-09);          /* 1/12! */
    p = fma(p, r, 2.505210838544172e-08);         /* 1/11! */
    p = fma(p, r, 2.755731922398589e-07);         /* 1/10! */
    p = fma(p, r, 2.7557319223985893e-06);        /* 1/9!  */
    p = fma(p, r, 2.48015873015873e-05);          /* 1/8!  */
    p = fma(p, r, 1.984126984126984e-04);         /* 1/7!  */
    p = fma(p, r, 1.388888888888889e-03);         /* 1/6!  */
    p = fma(p, r, 8.333333333333333e-03);         /* 1/5!  */
    p = fma(p, r, 4.1666666666666664e-02);        /* 1/4!  */
    p = fma(p, r, 1.6666666666666666e-01);        /* 1/3!  */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

/* log(x) for x > 0: the classic fdlibm/musl argument reduction and Lg1..Lg7
 * polynomial, every operation rounded separately (no fma). <1 ulp. */
static inline double nhp_det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u;
    memcpy(&u, &x, 8);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -INFINITY;      /* log(+-0) */
        if (hx >> 31) return NAN;                 /* log(<0)  */
        k -= 54;                                  /* subnormal: scale up */
        x *= 18014398509481984.0;                 /* 2^54 */
        memcpy(&u, &x, 8);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;                                 /* inf or nan */
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffu);
    memcpy(&x, &u, 8);
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

/* Philox4x32-10 (Salmon et al., SC'11), counter = (c0,c1,c2,c3), key = (k0,k1). */
static inline void nhp_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* The uniform stream contract: u(seed, step, event) in [0,1), 53 bits. */
static inline double nhp_uniform(uint64_t seed, uint64_t step, uint64_t event)
{
    uint32_t c[4] = { (uint32_t)event, (uint32_t)(event >> 32),
                      (uint32_t)step,  (uint32_t)(step >> 32) };
    nhp_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t bits = ((uint64_t)c[0] << 32) | c[1];
    return (double)(bits >> 11) * 1.1102230246251565e-16;   /* 2^-53 */
}

#endif
