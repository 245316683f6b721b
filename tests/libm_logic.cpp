// tests/libm_logic.cpp — raytracer_amd/csrc/rt_libm.h (the text the HIP kernels compile for rt_powf / rt_expf)
// built for the host and compared, bit for bit, with the C library of this box (glibc: the powf / expf the
// reference calls through std::pow / std::exp, raytracer.cpp:1356-1358,1428,1517).
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp -DRT_LIBM_HOST -Iraytracer_amd/csrc tests/libm_logic.cpp -lm
// (-mfma only so that fma() is one instruction; -ffp-contract=off: nothing but the explicit fma() calls fuses.)
// Operands: the shading ranges densely (Blinn-Phong: cos_alpha in [0, 1.003] to exponents 0.5 .. 2000; Beer:
// exp(-sigma * d) with arguments in [-60, 0]), every float bit pattern on a coarse stride for both functions, and
// the special cases (zeros, infinities, NaN, subnormals, negative bases with integer exponents, overflow and
// underflow thresholds).  Prints the number of operands and "total: 0 mismatches" when all agree.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "rt_libm.h"

static inline bool same(float a, float b) {
    uint32_t x, y;
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    if (x == y) return true;
    return std::isnan(a) && std::isnan(b);   // NaN payload / sign is not part of the contract
}
static inline uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33) ^ (uint32_t)(s >> 11); }
static inline float unit(uint64_t& s) { return (rng(s) >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    const long long scale = argc > 1 ? atoll(argv[1]) : 1;   // 1: ~1.3e8 operands
    unsigned long long n = 0, bad = 0;
    int printed = 0;
    auto report = [&](const char* what, float a, float b, float got, float want) {   // (called inside omp critical)
        if (printed++ < 20) printf("MISMATCH %s(%a, %a): got %a want %a\n", what, a, b, got, want);
    };
    // ---- expf: every 37th bit pattern of the whole float range
    {
        unsigned long long nb = 0, nn = 0;
#pragma omp parallel for reduction(+ : nb, nn) schedule(static)
        for (long long u = 0; u < (1ll << 32); u += 37) {
            float x;
            uint32_t b = (uint32_t)u;
            memcpy(&x, &b, 4);
            const float g = rt_glibc_expf(x), w = expf(x);
            nn++;
            if (!same(g, w)) { nb++; 
#pragma omp critical
                report("expf", x, 0, g, w); }
        }
        n += nn; bad += nb;
        printf("expf, whole range (stride 37): %llu operands, %llu mismatches\n", nn, nb);
    }
    // ---- expf: Beer's law arguments
    {
        unsigned long long nb = 0, nn = 0;
        const long long N = 40000000ll * scale;
#pragma omp parallel for reduction(+ : nb, nn) schedule(static)
        for (long long k = 0; k < N; k++) {
            uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
            const float sigma = unit(s) * (k % 3 == 0 ? 0.05f : k % 3 == 1 ? 1.0f : 20.0f), d = unit(s) * (k % 5 == 0 ? 30.0f : 3.0f);
            const float x = -sigma * d;
            const float g = rt_glibc_expf(x), w = expf(x);
            nn++;
            if (!same(g, w)) { nb++;
#pragma omp critical
                report("expf", x, 0, g, w); }
        }
        n += nn; bad += nb;
        printf("expf, Beer arguments: %llu operands, %llu mismatches\n", nn, nb);
    }
    // ---- powf: Blinn-Phong operands
    {
        unsigned long long nb = 0, nn = 0;
        const long long N = 60000000ll * scale;
        const float exps[] = {0.5f, 1.0f, 2.0f, 3.0f, 5.0f, 8.0f, 10.0f, 20.0f, 30.0f, 40.0f, 50.0f, 60.0f, 80.0f, 100.0f, 128.0f, 200.0f, 300.0f, 500.0f, 1000.0f, 2000.0f, 12.5f, 33.3f};
#pragma omp parallel for reduction(+ : nb, nn) schedule(static)
        for (long long k = 0; k < N; k++) {
            uint64_t s = 0xD1B54A32D192ED03ull * (uint64_t)(k + 1);
            float c = unit(s);
            if (k % 4 == 0) c = 1.0f - c * c * 0.01f;            // highlights: cos_alpha near 1
            if (k % 64 == 1) c = 1.0f + unit(s) * 0.003f;        // Q_rsqrt leaves |n|, |h| a little off 1
            const float e = (k % 7 == 0) ? unit(s) * 200.0f : exps[rng(s) % (sizeof(exps) / sizeof(exps[0]))];
            const float g = rt_glibc_powf(c, e), w = powf(c, e);
            nn++;
            if (!same(g, w)) { nb++;
#pragma omp critical
                report("powf", c, e, g, w); }
        }
        n += nn; bad += nb;
        printf("powf, Blinn-Phong operands: %llu operands, %llu mismatches\n", nn, nb);
    }
    // ---- powf: random bit patterns of both arguments (all branches), and a grid of special values
    {
        unsigned long long nb = 0, nn = 0;
        const long long N = 30000000ll * scale;
#pragma omp parallel for reduction(+ : nb, nn) schedule(static)
        for (long long k = 0; k < N; k++) {
            uint64_t s = 0xA0761D6478BD642Full * (uint64_t)(k + 1);
            uint32_t bx = rng(s), by = rng(s);
            if (k % 3 == 0) by = (by & 0x807fffffu) | ((120u + rng(s) % 16u) << 23);   // |y| around 2^-7 .. 2^8: finite results
            if (k % 5 == 0) { float yi = (float)((int)(rng(s) % 41u) - 20); memcpy(&by, &yi, 4); }   // integer exponents (negative bases)
            float x, y;
            memcpy(&x, &bx, 4);
            memcpy(&y, &by, 4);
            const float g = rt_glibc_powf(x, y), w = powf(x, y);
            nn++;
            if (!same(g, w)) { nb++;
#pragma omp critical
                report("powf", x, y, g, w); }
        }
        n += nn; bad += nb;
        printf("powf, random bit patterns: %llu operands, %llu mismatches\n", nn, nb);
        const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, -0.5f, 2.0f, -2.0f, 3.0f, -3.0f, INFINITY, -INFINITY, NAN, 0x1p-149f, -0x1p-149f, 0x1p-126f, 0x1.fffffep127f,
                            -0x1.fffffep127f, 0x1p-140f, 1.5f, 127.0f, 128.0f, -149.0f, -150.0f, 0x1p24f, 0x1p23f + 1.0f, 0.99999994f, 1.0000001f, 88.0f, -104.0f, 89.0f, -103.5f};
        nn = nb = 0;
        for (float a : sp)
            for (float b : sp) {
                const float g = rt_glibc_powf(a, b), w = powf(a, b);
                nn++;
                if (!same(g, w)) { nb++; report("powf", a, b, g, w); }
            }
        for (float a : sp) {
            const float g = rt_glibc_expf(a), w = expf(a);
            nn++;
            if (!same(g, w)) { nb++; report("expf", a, 0, g, w); }
        }
        n += nn; bad += nb;
        printf("special values: %llu operands, %llu mismatches\n", nn, nb);
    }
    printf("operands: %llu\ntotal: %llu mismatches\n", n, bad);
    return bad ? 1 : 0;
}
