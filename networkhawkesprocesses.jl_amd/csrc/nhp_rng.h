// Counter-based random variates for the device-side Gibbs draws (continuous: cont_sampler.hip, discrete:
// disc_gibbs.hip).
#pragma once
#include "nhp_internal.h"

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
    return sqrt(-2.0 * log(ua)) * cos(6.283185307179586 * ub);          // Box-Muller
}

// Gamma(shape, scale) by Marsaglia & Tsang (2000); shape < 1 via Gamma(shape+1)·U^(1/shape).
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
    const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double x = dev_normal(key, step, e, attempt++);
        double ua, ub;
        philox_2u(key, step, e, attempt++, &ua, &ub);
        const double t = 1.0 + c * x, v = t * t * t;
        if (v > 0.0 && log(ua) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * scale * boost;
        if (attempt > 200) return d * scale * boost;                       // unreachable in practice
    }
}

