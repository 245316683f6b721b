// rt_libm.h — std::pow(float,float) and std::exp(float) of the reference (raytracer.cpp:1356-1358, 1428, 1517),
// restated bit for bit.
//
// The reference calls the C library: glibc 2.35's powf / expf (libm.so.6 of the Ubuntu 22.04 toolchain the
// reference's committed renders reproduce with).  That is third-party arithmetic outside
// /root/reference, so its PUBLISHED algorithm is restated here: sysdeps/ieee754/flt-32/e_powf.c, e_expf.c,
// e_exp2f_data.c, e_powf_log2_data.c (the ARM "optimized routines" single-precision kernels: the argument is
// widened to double, a table-driven 2^(k/32) * polynomial(r) / 16-entry log2 table * polynomial, one final rounding
// to float).  Two facts of the BINARY matter for bit-exactness and were read off the x86-64 build:
//   * on a CPU with FMA, glibc's ifunc selects the variant of the same source compiled with -mfma, where gcc
//     contracts every `a * b + c` of the kernels into one fused operation — including, in expf,
//     kd = fma(InvLn2N, x, SHIFT) and r = fma(InvLn2N, x, -kd).  Those contractions are written out below as
//     explicit fma() calls (the device build uses -ffp-contract=off, so nothing else is fused);
//   * TOINT_INTRINSICS is 0 on x86-64: rounding to the nearest table step is done by adding and subtracting
//     the 1.5 * 2^52 shift constants.
// The difference to the non-FMA variant is a last-bit difference of a double before its rounding to float, i.e.
// invisible except about once in 2^29 calls; against the 1-2 ulp (float) of OCML's powf / expf, which this header
// replaces and which did flip a truncated byte (one pixel of a 328x184 frame in this repository's own tests).
//
// tests/libm_logic.cpp compiles this header for the host and compares it with the C library of the box it
// runs on over > 10^8 operands (the shading ranges densely, the whole float range sparsely, every special case).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef RT_LIBM_HOST
#include <math.h>
#define RT_LIBM_FN static inline
#define RT_LIBM_CONST static const
#define rt_fma_f64(a, b, c) fma((a), (b), (c))
static inline uint32_t rt_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t rt_f64_bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double rt_bits_f64(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline float rt_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#else
#include <hip/hip_runtime.h>
#define RT_LIBM_FN __device__ __forceinline__
#define RT_LIBM_CONST __device__ const
#define rt_fma_f64(a, b, c) __builtin_fma((a), (b), (c))
__device__ __forceinline__ uint32_t rt_f32_bits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint64_t rt_f64_bits(double f) { return (uint64_t)__double_as_longlong(f); }
__device__ __forceinline__ double rt_bits_f64(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ float rt_bits_f32(uint32_t u) { return __uint_as_float(u); }
#endif

// e_exp2f_data.c: tab[i] = bits(2^(i/32)) - (i << 47)
RT_LIBM_CONST uint64_t rt_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
// e_powf_log2_data.c: {invc, logc} per sixteenth of [2^-0.5 * ..., ...): log2(x) = log1p(x * invc - 1) / ln2 + logc + k
RT_LIBM_CONST double rt_powf_log2_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1p+0, 0x0p+0},                              {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};

// math_errf.c without errno: the values only
RT_LIBM_FN float rt_math_oflowf(uint32_t sign) { const float y = 0x1p97f; return (sign ? -y : y) * y; }
RT_LIBM_FN float rt_math_uflowf(uint32_t sign) { const float y = 0x1p-95f; return (sign ? -y : y) * y; }
RT_LIBM_FN float rt_math_may_uflowf(uint32_t sign) { const float y = 0x1.4p-75f; return (sign ? -y : y) * y; }
RT_LIBM_FN float rt_math_invalidf(float x) { return (x - x) / (x - x); }

// e_expf.c (FMA variant, see the header)
RT_LIBM_FN float rt_glibc_expf(float x) {
    const double xd = (double)x;
    const uint32_t abstop = (rt_f32_bits(x) >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {   // top12(88.0f): |x| >= 88 or NaN
        if (rt_f32_bits(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return rt_math_oflowf(0);    // x > log(0x1p128)
        if (x < -0x1.9fe368p6f) return rt_math_uflowf(0);   // x < log(0x1p-150)
        if (x < -0x1.9d1d9ep6f) return rt_math_may_uflowf(0);
    }
    const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
    // x * N / ln2 = k + r, r in [-1/2, 1/2]
    double kd = rt_fma_f64(InvLn2N, xd, SHIFT);
    const uint64_t ki = rt_f64_bits(kd);
    kd -= SHIFT;
    const double r = rt_fma_f64(InvLn2N, xd, -kd);
    // exp(x) = 2^(k/N) * 2^(r/N) ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
    const uint64_t t = rt_exp2f_tab[ki & 31u] + (ki << 47);
    const double s = rt_bits_f64(t);
    const double z = rt_fma_f64(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    const double r2 = r * r;
    double y = rt_fma_f64(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = rt_fma_f64(z, r2, y);
    y = y * s;
    return (float)y;
}

// e_powf.c: 1 if the float with bits iy is an odd integer, 2 if an even integer, 0 otherwise
RT_LIBM_FN int rt_powf_checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
RT_LIBM_FN bool rt_powf_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
RT_LIBM_FN bool rt_issignalingf(uint32_t ix) { return 2u * (ix ^ 0x00400000u) > 2u * 0x7fc00000u; }   // math_config.h: issignalingf_inline

// e_powf.c (FMA variant)
RT_LIBM_FN float rt_glibc_powf(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = rt_f32_bits(x);
    const uint32_t iy = rt_f32_bits(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || rt_powf_zeroinfnan(iy)) {
        // either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
        if (rt_powf_zeroinfnan(iy)) {
            if (2u * iy == 0u) return rt_issignalingf(ix) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return rt_issignalingf(iy) ? x + y : 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   // |x| < 1 and y = inf, or |x| > 1 and y = -inf
            return y * y;
        }
        if (rt_powf_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && rt_powf_checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        // x and y are non-zero finite
        if (ix & 0x80000000u) {
            const int yint = rt_powf_checkint(iy);
            if (yint == 0) return rt_math_invalidf(x);
            if (yint == 1) sign_bias = 1u << (5 + 11);                       // SIGN_BIAS
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {   // normalise a subnormal x
            ix = rt_f32_bits(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline: x = 2^k z, z in [OFF, 2 OFF), 16 subintervals
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = rt_powf_log2_tab[i][0], logc = rt_powf_log2_tab[i][1];
    const double z = (double)rt_bits_f32(iz);
    const double r = rt_fma_f64(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const double r2 = r * r;
    double yy = rt_fma_f64(A0, r, A1);
    const double p = rt_fma_f64(A2, r, A3);
    const double r4 = r2 * r2;
    double q = rt_fma_f64(A4, r, y0);
    q = rt_fma_f64(p, r2, q);
    yy = rt_fma_f64(yy, r4, q);
    const double logx = yy;
    const double ylogx = (double)y * logx;   // cannot overflow: y is single precision
    if ((rt_f64_bits(ylogx) >> 47 & 0xffffu) >= (0x405f800000000000ull >> 47)) {   // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return rt_math_oflowf(sign_bias);
        // (glibc re-checks 0x1.fffffffa3aae2p+6 for the directed rounding modes: nothing to do in round-to-nearest)
        if (ylogx <= -150.0) return rt_math_uflowf(sign_bias);
        if (ylogx < -149.0) return rt_math_may_uflowf(sign_bias);
    }
    // exp2_inline: x = k/N + r, r in [-1/(2N), 1/(2N)]
    const double SHIFT = 0x1.8p+47;
    double kd = ylogx + SHIFT;
    const uint64_t ki = rt_f64_bits(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    const uint64_t t = rt_exp2f_tab[ki & 31u] + ((ki + sign_bias) << 47);
    const double s = rt_bits_f64(t);
    const double zz = rt_fma_f64(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double out = rt_fma_f64(0x1.62e42ff0c52d6p-1, rr, 1.0);
    out = rt_fma_f64(zz, rr2, out);
    out = out * s;
    return (float)out;
}
