// Host check of raytracer_amd/csrc/rt_mt.h (the text the HIP kernels compile) against libstdc++ itself: what the
// reference's pixel loop does with its per-pixel generator (raytracer.cpp:44-69) is std::mt19937(seed), one
// std::shuffle of the sample indices, and std::uniform_real_distribution<float>(0, 1) draws.
#include <hip/hip_runtime.h>   // tests/shim
#include "../raytracer_amd/csrc/rt_mt.h"
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

static long mismatches = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (mismatches < 20) { printf(__VA_ARGS__); printf("\n"); } mismatches++; } } while (0)

int main() {
    std::vector<uint32_t> seeds = {0u, 1u, 2u, 5489u, 12345u, 1920u * 1080u - 1u, 7680u * 4320u - 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu};
    std::mt19937 pick(99);
    for (int k = 0; k < 150; k++) seeds.push_back((uint32_t)pick());
    const int Ns[] = {1, 2, 3, 4, 5, 9, 16, 25, 36, 64, 100, 121, 400, 900};
    long checked = 0;
    for (uint32_t seed : seeds) {
        // (a) raw outputs across five blocks, with the state columns strided as on the device
        {
            const size_t stride = 3;
            std::vector<uint32_t> mem(624 * stride, 0xDEADBEEFu);
            MtGlobal g;
            g.mt = mem.data() + 1;
            g.stride = stride;
            g.seed(seed);
            std::mt19937 ref(seed);
            for (int i = 0; i < 3200; i++) {
                const uint32_t a = g.next(), b = (uint32_t)ref();
                CHECK(a == b, "seed %u draw %d: %08x != %08x", seed, i, a, b);
                checked++;
                if (i % 97 == 13) {           // the generator's position survives being parked in three words (rt_ordered.h)
                    MtGlobal h;
                    h.mt = g.mt; h.stride = g.stride;
                    h.unpack(g.packed_idx(), g.lo, g.hi);
                    g = h;
                }
            }
            uint32_t f5[5];
            mt_first5(seed, f5);
            std::mt19937 r5(seed);
            for (int i = 0; i < 5; i++) { const uint32_t b = (uint32_t)r5(); CHECK(f5[i] == b, "seed %u first5[%d]", seed, i); }
        }
        // (b) shuffle of N sample indices, then canonical floats, as the pixel loop draws them
        for (int N : Ns) {
            std::vector<uint32_t> mem(624), perm((size_t)N);
            MtGlobal g;
            g.mt = mem.data();
            g.stride = 1;
            g.seed(seed);
            mt_shuffle_identity(g, perm.data(), 1, N);
            std::mt19937 ref(seed);
            std::vector<int> idx((size_t)N);
            std::iota(idx.begin(), idx.end(), 0);
            std::shuffle(idx.begin(), idx.end(), ref);
            for (int i = 0; i < N; i++) CHECK((int)perm[(size_t)i] == idx[(size_t)i], "seed %u N %d: perm[%d] = %u, std::shuffle gives %d", seed, N, i, perm[(size_t)i], idx[(size_t)i]);
            std::uniform_real_distribution<float> uni(0.0f, 1.0f);
            for (int i = 0; i < 5 * N + 40; i++) {
                const float a = g.uniform(), b = uni(ref);
                CHECK(a == b, "seed %u N %d: float %d: %.9g != %.9g", seed, N, i, a, b);
                checked++;
            }
        }
    }
    // (c) the canonical float at the edges of the 32-bit range (the clamp below 1)
    for (uint32_t u : {0u, 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFF00u, 0xFFFFFF7Fu, 0xFFFFFF80u, 0xFFFFFFFEu, 0xFFFFFFFFu}) {
        struct One { uint32_t v; using result_type = uint32_t; static constexpr uint32_t min() { return 0; } static constexpr uint32_t max() { return 0xFFFFFFFFu; } uint32_t operator()() { return v; } } one{u};
        const float b = std::generate_canonical<float, 24>(one);
        CHECK(u32_to_unit(u) == b, "u32_to_unit(%08x) = %.9g, generate_canonical %.9g", u, u32_to_unit(u), b);
    }
    printf("checked %ld values; total: %ld mismatches\n", checked, mismatches);
    return mismatches ? 1 : 0;
}
