// Counter-based random variates for the device-side Gibbs draws (continuous: cont_sampler.hip, discrete:
// disc_gibbs.hip).
#pragma once
#include "nhp_internal.h"
#include "nhp_math.h"

// Lean elementary functions for the random variates (the library's correctly-rounded-ish log / cos / division with
// their full-range argument reduction made k_gibbs_draw a 1700-instruction kernel).  Accurate to a few 1e-15, which is
// all a random draw needs; none of this is on a parity path.
__device__ __forceinline__ double rng_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}
__device__ __forceinline__ double rng_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}
// cos(2π·c/2^32): quadrant from the top two bits, fold the quarter turn to [0, π/4], Taylor sine / cosine there
__device__ __forceinline__ double rng_cos_turn(uint32_t c)
{
    const uint32_t q = c >> 30;
    uint32_t k = c & 0x3FFFFFFFu;
    bool want_sin = q & 1u;
    if (k > 0x20000000u) { k = 0x40000000u - k; want_sin = !want_sin; }
    const double y = (double)k * 1.4629180792671596e-9;          // (π/2) / 2^30
    const double z = y * y;
    const double sn = y * (1.0 + z * (-1.6666666666666666e-1 + z * (8.3333333333333332e-3 + z * (-1.9841269841269841e-4 +
                      z * (2.7557319223985893e-6 + z * (-2.5052108385441720e-8 + z * (1.6059043836821613e-10 + z * -7.6471637318198164e-13)))))));
    const double cs = 1.0 + z * (-0.5 + z * (4.1666666666666664e-2 + z * (-1.3888888888888889e-3 + z * (2.4801587301587302e-5 +
                      z * (-2.7557319223985888e-7 + z * (2.0876756987868100e-9 + z * (-1.1470745597729725e-11 + z * 4.7794773323873853e-14)))))));
    const double v = want_sin ? sn : cs;
    return (q == 1u || q == 2u) ? -v : v;
}

// ---- device-side conjugate draws (reference resample! bodies: src/baselines.jl:72-77,
// src/weights.jl:59-64, src/impulses.jl:68-73,204-214).  Counter-based: element e of draw family
// `fam` at chain step `step` consumes Philox counters (e, attempt) under key (seed ^ fam-constant,
// step), so a chain is reproducible on the device and independent of launch geometry.  Julia's
// samplers cannot be matched bit for bit ([3P] Distributions / Random); parity is distributional.
__device__ __forceinline__ void philox_2u(uint64_t key, uint64_t step, uint64_t e, uint32_t attempt, double *ua, double *ub)
{
    uint32_t c0 = (uint32_t)e, c1 = (uint32_t)(e >> 32) ^ (attempt << 8), c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // (0,1]: never 0, so log() below is finite
    *ua = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 1.0) * 1.1102230246251565e-16;
    *ub = ((double)((((uint64_t)c2 << 32) | c3) >> 11) + 1.0) * 1.1102230246251565e-16;
}

__device__ __forceinline__ double dev_normal(uint64_t key, uint64_t step, uint64_t e, uint32_t attempt)
{
    double ua, ub;
    philox_2u(key, step, e, attempt, &ua, &ub);
    return sqrt(-2.0 * nhp_log(ua)) * rng_cos_turn((uint32_t)(ub * 4294967296.0));          // Box-Muller
}

// One Philox block per Marsaglia-Tsang attempt: 53 bits for the Box-Muller radius (the normal's tail), 32 for its
// angle, 32 for the acceptance uniform.
__device__ __forceinline__ void philox_attempt(uint64_t key, uint64_t step, uint64_t e, uint32_t attempt, double *x, double *u)
{
    uint32_t c0 = (uint32_t)e, c1 = (uint32_t)(e >> 32) ^ (attempt << 8), c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const double ua = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 1.0) * 1.1102230246251565e-16;     // (0,1]
    *x = sqrt(-2.0 * nhp_log(ua)) * rng_cos_turn(c2);
    *u = ((double)c3 + 0.5) * 2.3283064365386963e-10;                                                    // (0,1)
}

// Gamma(shape, scale) by Marsaglia & Tsang (2000), with their squeeze u < 1 - 0.0331 x^4 ahead of the logarithms;
// shape < 1 via Gamma(shape+1)·U^(1/shape).
__device__ inline double dev_gamma(double shape, double scale, uint64_t key, uint64_t step, uint64_t e)
{
    double boost = 1.0;
    uint32_t attempt = 0;
    if (shape < 1.0) {
        double ua, ub;
        philox_2u(key, step, e, attempt++, &ua, &ub);
        boost = pow(ua, 1.0 / shape);
        shape += 1.0;
    }
    const double d = shape - 1.0 / 3.0, c = rng_rsqrt(9.0 * d);
    for (;;) {
        double x, u;
        philox_attempt(key, step, e, attempt++, &x, &u);
        const double t = 1.0 + c * x, v = t * t * t, x2 = x * x;
        if (v > 0.0 && (u < 1.0 - 0.0331 * x2 * x2 || nhp_log(u) < 0.5 * x2 + d - d * v + d * nhp_log(v))) return d * v * scale * boost;
        if (attempt > 200) return d * scale * boost;                       // unreachable in practice
    }
}
