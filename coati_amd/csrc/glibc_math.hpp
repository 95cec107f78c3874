// Bit-exact device restatements of the three libm functions the reference's log-semiring path
// calls, so that the Forward matrices and the sampled log-weights are IDENTICAL to the CPU's:
//
//   expf    log1p_exp (src/include/coati/utils.hpp:134-146), sample_mdi/sample_mi (align_pair.cc:336-385)
//   log1pf  log1p_exp
//   logf    sample_mdi/sample_mi
//
// The algorithm is a third-party dependency that is not under /root/reference: GNU libc 2.35 libm
// (Ubuntu GLIBC 2.35-0ubuntu3.11, x86_64), the libm of the build container and of the GPU box:
//   expf, logf   sysdeps/ieee754/flt-32/e_expf.c, e_logf.c -- the ARM "optimized routines" single
//                precision kernels (Szabolcs Nagy): double-precision polynomial around a 32- / 16-entry
//                table; on x86_64 CPUs with FMA glibc dispatches to the FMA-contracted build, which is
//                what every machine in this project has.  The contraction pattern below (which
//                products are fused) is the one that build has.
//   log1pf       sysdeps/ieee754/flt-32/s_log1pf.c -- the fdlibm (Sun) algorithm in float; no FMA
//                variant exists.
// Licence of what is restated: glibc is LGPL-2.1-or-later; e_expf.c / e_logf.c carry the ARM optimized-routines
// notice (Copyright (C) 2017-2022 Free Software Foundation / Arm Ltd., LGPL-2.1-or-later in glibc, MIT OR
// Apache-2.0 WITH LLVM-exception upstream), s_log1pf.c the fdlibm notice ("Copyright (C) 1993 by Sun
// Microsystems, Inc. ... Permission to use, copy, modify, and distribute this software is freely granted,
// provided that this notice is preserved").  What is here is a re-derivation of the published ALGORITHMS
// (polynomial coefficients and table values are mathematical constants of those algorithms) written for the
// device, not a copy of glibc's source text; the constants' provenance is the files named above.
// Parity is pinned, not assumed: tools/libm_check.cc compares these restatements (compiled for the
// host) with the container's libm on EVERY float of the ranges the path can produce -- expf on
// [-104, 0] (1 120 927 745 inputs), log1pf on [0, 1] (1 065 353 217), logf on [2^-126, 4]
// (1 073 741 825) -- 0 mismatches; tests/test_gpu_math.py compares the DEVICE code with the host libm
// on dense samples of the same ranges.
#ifndef COATI_HIP_GLIBC_MATH_HPP
#define COATI_HIP_GLIBC_MATH_HPP

#include <cstdint>

#ifndef COATI_MATH_FN
#define COATI_MATH_FN __device__ __forceinline__
#endif

namespace coati_hip_detail {
namespace libm {

COATI_MATH_FN uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
COATI_MATH_FN float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
COATI_MATH_FN uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
COATI_MATH_FN double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// tab[i] = asuint64(2^(i/32)) - (i << 47)
#define COATI_EXP2F_TABLE                                                                             \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,     \
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,     \
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,     \
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,     \
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,     \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,     \
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,     \
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull

// expf for x <= 0 (the only arguments the path produces: -|a-b| and lx - max <= 0).
// `tab` is the 32-entry table above (callers keep it in LDS or constant memory).
COATI_MATH_FN float expf_nonpos(float x, const uint64_t* tab) {
    constexpr double kInvLn2N = 0x1.71547652b82fep+0 * 32, kShift = 0x1.8p+52;
    constexpr double kC0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, kC1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
                     kC2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double xd = static_cast<double>(x);
    // x*N/ln2 = k + r, r in [-1/2, 1/2]: the FMA build never materialises the product
    double kd = __builtin_fma(xd, kInvLn2N, kShift);
    const uint64_t ki = d2u(kd);
    kd -= kShift;
    const double r = __builtin_fma(xd, kInvLn2N, -kd);
    // exp(x) = 2^(k/N) * 2^(r/N) ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
    const double s = u2d(tab[ki % 32] + (ki << (52 - 5)));
    const double z = __builtin_fma(kC0, r, kC1);
    const double r2 = r * r;
    double y = __builtin_fma(r, kC2, 1.0);
    y = __builtin_fma(z, r2, y);
    // x < log(2^-150) underflows to +0 (also -inf, -FLT_MAX; whatever the lines above made of those
    // is discarded -- a select, not a branch, so wavefronts stay converged)
    return x < -0x1.9fe368p6f ? 0.0f : static_cast<float>(y * s);
}

// log1pf for 0 <= x <= 1 (x = expf(y), -16 < y <= 0).  fdlibm: 1+x = 2^k (1+f), log(1+f) by a
// degree-7 polynomial in s = f/(2+f); c corrects the rounding of 1+x.
COATI_MATH_FN float log1pf_unit(float x) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f,
                    Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                    Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    float f = 0.0f, c = 0.0f, u;
    const int32_t hx = static_cast<int32_t>(f2u(x));
    int32_t k = 1, hu = 0;
    if(hx < 0x3ed413d7) {          // x < 0.41422
        if(hx < 0x31000000) {      // x < 2^-29
            if(two25 + x > 0.0f && hx < 0x24800000) return x;  // x < 2^-54
            return x - x * x * 0.5f;
        }
        k = 0;                     // sqrt(2)/2 < 1+x < sqrt(2): no scaling (x > 0 here)
        f = x;
        hu = 1;
    }
    if(k != 0) {
        u = 1.0f + x;
        hu = static_cast<int32_t>(f2u(u));
        k = (hu >> 23) - 127;
        c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);  // correction term
        c /= u;
        hu &= 0x007fffff;
        if(hu < 0x3504f7) {
            u = u2f(static_cast<uint32_t>(hu) | 0x3f800000u);  // normalise u
        } else {
            k += 1;
            u = u2f(static_cast<uint32_t>(hu) | 0x3f000000u);  // normalise u/2
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    const float hfsq = 0.5f * f * f;
    const float kf = static_cast<float>(k);
    if(hu == 0) {  // |f| < 2^-20
        if(f == 0.0f) {
            if(k == 0) return 0.0f;
            c += kf * ln2_lo;
            return kf * ln2_hi + c;
        }
        const float R = hfsq * (1.0f - 0.66666666666666666f * f);
        if(k == 0) return f - R;
        return kf * ln2_hi - ((R - (kf * ln2_lo + c)) - f);
    }
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if(k == 0) return f - (hfsq - s * (hfsq + R));
    return kf * ln2_hi - ((hfsq - (s * (hfsq + R) + (kf * ln2_lo + c))) - f);
}

// ---- log1pf on [2^-29, 1], straight-line -------------------------------------------------------
// The Forward fill only ever asks for log1pf(e), e = expf(y) with -16 < y <= 0, i.e. e >= 1.1e-7.
// On that range log1pf_unit above takes one of two main routes (k = 0: f = x; k = 1: f = u/2 - 1),
// which are evaluated side by side here and selected, so a wavefront never diverges; the handful of
// inputs on other routes (u = 1+x rounding to 2, or to within 3 ulp of it) branch to log1pf_near2.
// The two divisions are replaced by reciprocal + Newton + Markstein correction (operands are
// normal and no intermediate can over/underflow: f/(2+f) with |f| in [2^-29, 0.42]; c/u with c zero
// or +-2^-24, +-2^-25, where RN(c/u) = c * RN(1/u) exactly).  Same bits as log1pf_unit -- and so as
// glibc -- on EVERY float of [2^-29, 1]: tools/libm_check.cc (host build, 1/d for the estimate) and
// tools/libm_device_check.py (the device code with v_rcp_f32) both sweep the whole range.
COATI_MATH_FN float rcp_estimate(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(d);  // v_rcp_f32, 1 ulp
#else
    return 1.0f / d;
#endif
}
COATI_MATH_FN float recip_rn(float d) {
    const float y0 = rcp_estimate(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
COATI_MATH_FN float div_rn(float n, float d) {
    const float y = recip_rn(d);
    const float q = n * y;
    const float r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, y, q);
}

// x in [0.41422, 1] whose u = fl(1+x) has a zero mantissa (u = 2) or lies within 3 ulp below 2:
// log1pf_unit's |f| < 2^-20 routes.
COATI_MATH_FN float log1pf_near2(float x, float u, uint32_t mant) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    if(mant == 0) {  // u == 2: k = 1, f = 0
        const float c = (1.0f - (u - x)) / u;
        return ln2_hi + (c + ln2_lo);
    }
    const float c = (x - (u - 1.0f)) / u;
    const float f = u * 0.5f - 1.0f;
    const float hfsq = 0.5f * f * f;
    const float R = hfsq * (1.0f - 0.66666666666666666f * f);
    return ln2_hi - ((R - (ln2_lo + c)) - f);
}

COATI_MATH_FN float log1pf_mid(float x) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, Lp1 = 6.6666668653e-01f,
                    Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f,
                    Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    const bool small = x < u2f(0x3ed413d7u);  // 1+x < sqrt(2): k = 0, f = x
    const float u = 1.0f + x;
    const uint32_t mant = f2u(u) & 0x007fffffu;
    const bool halve = !small && mant >= 0x3504f7u;  // k = 1, f = u/2 - 1 (else k = 0, f = u - 1)
    const float f = small ? x : __builtin_fmaf(u, halve ? 0.5f : 1.0f, -1.0f);  // (u/2 is exact)
    const float c = (x - (u - 1.0f)) * recip_rn(u);  // rounding error of 1+x, relative
    const float hfsq = 0.5f * f * f;
    const float s = div_rn(f, 2.0f + f);
    const float z = s * s;
    const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    const float t = s * (hfsq + R);
    const float r0 = f - (hfsq - t);
    const float r1 = ln2_hi - ((hfsq - (t + (ln2_lo + c))) - f);
    float res = halve ? r1 : r0;
    if(__builtin_expect(!small && mant - 1u >= 0x7ffffcu, 0)) res = log1pf_near2(x, u, mant);
    return res;
}

// The k = 0 route alone (x < 0.41422: 1 + x < sqrt(2), f = x): what log1pf_mid computes for such x, without the k = 1 side
// (u = 1 + x, its mantissa test, the correction c / u, the second result) that log1pf_mid evaluates for every lane and then
// discards.  For a wavefront whose lanes ALL take this route (or whose result is not used: y <= -16); round 4,
// tools/forward_y_hist.py: a quarter to two fifths of the Forward's `plus` operations at 8 columns per lane.
COATI_MATH_FN float log1pf_small(float x) {
    constexpr float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f,
                    Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    const float f = x;
    const float hfsq = 0.5f * f * f;
    const float s = div_rn(f, 2.0f + f);
    const float z = s * s;
    const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    const float t = s * (hfsq + R);
    return f - (hfsq - t);
}

// Two independent log1pf_mid side by side, as ONE straight-line block (the rare |f| < 2^-20 routes are redone after both):
// the exact Forward cell has two such dependent chains at a time that do not depend on each other (the M and the D sums),
// and a chain of ~40 fp64 / reciprocal / fp32 steps issues no faster than its latencies allow.  Same operations on the
// same values as log1pf_mid, so the same bits.
COATI_MATH_FN void log1pf_mid_x2(float x0, float x1, float& out0, float& out1) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, Lp1 = 6.6666668653e-01f,
                    Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f, Lp5 = 1.8183572590e-01f,
                    Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    const float x[2] = {x0, x1};
    float res[2], u[2];
    uint32_t mant[2];
    bool rare[2];
#pragma unroll
    for(int q = 0; q < 2; ++q) {
        const bool small = x[q] < u2f(0x3ed413d7u);
        u[q] = 1.0f + x[q];
        mant[q] = f2u(u[q]) & 0x007fffffu;
        const bool halve = !small && mant[q] >= 0x3504f7u;
        const float f = small ? x[q] : __builtin_fmaf(u[q], halve ? 0.5f : 1.0f, -1.0f);
        const float c = (x[q] - (u[q] - 1.0f)) * recip_rn(u[q]);
        const float hfsq = 0.5f * f * f;
        const float s = div_rn(f, 2.0f + f);
        const float z = s * s;
        const float R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
        const float t = s * (hfsq + R);
        const float r0 = f - (hfsq - t);
        const float r1 = ln2_hi - ((hfsq - (t + (ln2_lo + c))) - f);
        res[q] = halve ? r1 : r0;
        rare[q] = !small && mant[q] - 1u >= 0x7ffffcu;
    }
    if(__builtin_expect(rare[0], 0)) res[0] = log1pf_near2(x[0], u[0], mant[0]);
    if(__builtin_expect(rare[1], 0)) res[1] = log1pf_near2(x[1], u[1], mant[1]);
    out0 = res[0];
    out1 = res[1];
}

// logf for normal positive x (x = sum of up to three expf values in (0, 3]).
COATI_MATH_FN float logf_pos(float x) {
    // {1/c, log(c)} for the 16 sub-intervals of [sqrt(2)/2, sqrt(2)) * 2^k
    constexpr double kInvc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                                  0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                                  0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                                  0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    constexpr double kLogc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                                  -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                                  -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                                  0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    constexpr double kLn2 = 0x1.62e42fefa39efp-1, kA0 = -0x1.00ea348b88334p-2, kA1 = 0x1.5575b0be00b6ap-2,
                     kA2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = f2u(x);
    if(ix == 0x3f800000u) return 0.0f;
    if(ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // subnormal, zero, negative, inf, nan
        if(ix * 2 == 0) return -__builtin_inff();
        if(ix == 0x7f800000u) return x;
        if((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = f2u(x * 0x1p23f) - (23u << 23);  // normalise a subnormal
    }
    // x = 2^k z, z in [0x3f330000, 2 * 0x3f330000): the interval index comes from the top mantissa bits
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = static_cast<int>((tmp >> (23 - 4)) % 16);
    const int k = static_cast<int32_t>(tmp) >> 23;
    const double z = static_cast<double>(u2f(ix - (tmp & 0xff800000u)));
    // log(x) = log1p(z/c - 1) + log(c) + k ln2
    const double r = __builtin_fma(z, kInvc[i], -1.0);
    const double y0 = __builtin_fma(static_cast<double>(k), kLn2, kLogc[i]);
    const double r2 = r * r;
    double y = __builtin_fma(kA1, r, kA2);
    y = __builtin_fma(kA0, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return static_cast<float>(y);
}

}  // namespace libm
}  // namespace coati_hip_detail
#endif
