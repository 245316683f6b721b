// rt_mt.h — std::mt19937, std::uniform_real_distribution<float> and std::shuffle as libstdc++ 11 runs them for the
// reference's per-pixel generator (raytracer.cpp:44-69; bits/random.tcc, bits/uniform_int_dist.h, bits/stl_algo.h),
// restated for the device.  Plain integer and float arithmetic, no wave-level code: tests/mt_logic.cpp compiles this
// very text for the host (over tests/shim/hip/hip_runtime.h) and compares it with libstdc++ itself — outputs across
// several 624-word blocks, the canonical floats, the shuffles, and the generator's position after each.
#pragma once
#include <cstddef>
#include <cstdint>

// ------------------------------------------------------ libstdc++ RNG restated
// std::mt19937 / std::shuffle / std::uniform_real_distribution<float> as used at
// raytracer.cpp:44-69 (libstdc++ 11: bits/random.tcc, bits/stl_algo.h,
// bits/uniform_int_dist.h).
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// generate_canonical<float,24>: float(u32)/2^32, clamped below 1
__device__ __forceinline__ float u32_to_unit(uint32_t u) {
    float r = (float)u * 2.3283064365386963e-10f;   // exact 2^-32 scale == division by 4294967296.0f
    if (r >= 1.0f) r = 0x1.fffffep-1f;
    return r;
}
// The first five outputs of mt19937(seed) need state words 0..5 and 397..401 of
// the seeding recurrence only: output k = temper(s[k+397] ^ mix(s[k], s[k+1])).
__device__ __forceinline__ void mt_first5(uint32_t seed, uint32_t out[5]) {
#define RT_MT_STEP(prev, i) (1812433253u * ((prev) ^ ((prev) >> 30)) + (uint32_t)(i))
    const uint32_t s0 = seed;
    const uint32_t s1 = RT_MT_STEP(s0, 1), s2 = RT_MT_STEP(s1, 2), s3 = RT_MT_STEP(s2, 3), s4 = RT_MT_STEP(s3, 4),
                   s5 = RT_MT_STEP(s4, 5);
    uint32_t s = s5;
#pragma unroll 4
    for (uint32_t i = 6; i <= 396; i++) s = RT_MT_STEP(s, i);
    const uint32_t h0 = RT_MT_STEP(s, 397), h1 = RT_MT_STEP(h0, 398), h2 = RT_MT_STEP(h1, 399),
                   h3 = RT_MT_STEP(h2, 400), h4 = RT_MT_STEP(h3, 401);
#undef RT_MT_STEP
    out[0] = mt_temper(mt_mix(s0, s1, h0));
    out[1] = mt_temper(mt_mix(s1, s2, h1));
    out[2] = mt_temper(mt_mix(s2, s3, h2));
    out[3] = mt_temper(mt_mix(s3, s4, h3));
    out[4] = mt_temper(mt_mix(s4, s5, h4));
}

// Full generator with its 624-word state in global scratch, one column per
// thread (word k of thread t at mt[k*stride + t]: coalesced while lanes are in
// step).  Used when num_samples > 1.
struct MtGlobal {
    uint32_t* mt;
    size_t stride;
    uint32_t idx;
    // The FIRST block after seeding is made without the seeded state ever being written: seed word i is a chain
    // s_i = 1812433253 * (s_{i-1} ^ (s_{i-1} >> 30)) + i, and making word k of the first block needs the seed words k,
    // k + 1 and (k < 227) k + 397 — two running values of that chain, `lo` at k and `hi` at k + 397 — or (k >= 227)
    // the new word k - 227, which is in memory.  seed() is then 397 multiply-adds in registers instead of 624 strided
    // stores a lane (100 us for a wave of the depth-first pixel loop whenever one of its lanes starts a pixel).
    uint32_t lo, hi;
    bool first;
    __device__ __forceinline__ uint32_t& w(uint32_t k) { return mt[(size_t)k * stride]; }
    static __device__ __forceinline__ uint32_t seed_step(uint32_t s, uint32_t i) { return 1812433253u * (s ^ (s >> 30)) + i; }
    __device__ void seed(uint32_t s) {
        lo = s;
        uint32_t h = s;
        for (uint32_t i = 1; i <= 397; i++) h = seed_step(h, i);
        hi = h;
        first = true;
        idx = 0;
    }
    __device__ __forceinline__ uint32_t packed_idx() const { return idx | (first ? 0x80000000u : 0u); }
    __device__ __forceinline__ void unpack(uint32_t p, uint32_t l, uint32_t h) { idx = p & 0x7FFFFFFFu; first = (p >> 31) != 0u; lo = l; hi = h; }
    // std::mersenne_twister_engine regenerates all 624 words when the block is used up (_M_gen_rand) and then hands them
    // out one by one.  Here word k of the new block is made when it is handed out: it needs the old words k, k + 1 and
    // (k < 227) k + 397, none of them overwritten yet, or (k >= 227) the new word k - 227, made 227 draws ago — the same
    // values in the same order, without a 624-word loop that stalls the 63 other lanes of a wave whenever one lane's
    // block runs out (lanes of the depth-first pixel loop use up their blocks at different moments: 95 us per stall).
    __device__ __forceinline__ uint32_t next() {
        if (first) {
            const uint32_t k = idx;                                   // 0 .. 623
            const uint32_t s1 = seed_step(lo, k + 1);                 // seed word k + 1
            const uint32_t b = (k == 623) ? w(0) : s1;                // (word 0 of this block is made already)
            const uint32_t m = (k < 227) ? hi : w(k - 227);
            const uint32_t v = mt_mix(lo, b, m);
            w(k) = v;
            lo = s1;
            if (k < 227) hi = seed_step(hi, k + 398);
            idx = k + 1;
            first = idx < 624;
            return mt_temper(v);
        }
        if (idx >= 624) idx = 0;
        const uint32_t k = idx;
        const uint32_t k1 = (k + 1 == 624) ? 0 : k + 1;
        const uint32_t km = (k + 397 >= 624) ? k + 397 - 624 : k + 397;
        const uint32_t v = mt_mix(w(k), w(k1), w(km));
        w(k) = v;
        idx = k + 1;
        return mt_temper(v);
    }
    __device__ __forceinline__ float uniform() { return u32_to_unit(next()); }
    // uniform_int_distribution::_S_nd<uint64_t> (Lemire)
    __device__ uint32_t lemire(uint32_t range) {
        unsigned long long product = (unsigned long long)next() * range;
        uint32_t low = (uint32_t)product;
        if (low < range) {
            uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                product = (unsigned long long)next() * range;
                low = (uint32_t)product;
            }
        }
        return (uint32_t)(product >> 32);
    }
};

// std::shuffle(v.begin(), v.end(), gen) on the identity permutation 0 .. N-1 (bits/stl_algo.h, the path that draws one
// number for two positions; N * N <= 2^32 - 1 is checked on the host).  perm[i * stride] is element i.
template <class Gen>
__device__ __forceinline__ void mt_shuffle_identity(Gen& gen, uint32_t* perm, size_t stride, int N) {
    for (int k = 0; k < N; k++) perm[(size_t)k * stride] = (uint32_t)k;
    int i = 1;
    if ((N % 2) == 0) {
        uint32_t dpos = gen.lemire(2);
        uint32_t a = perm[(size_t)i * stride], b = perm[(size_t)dpos * stride];
        perm[(size_t)i * stride] = b; perm[(size_t)dpos * stride] = a;
        i++;
    }
    while (i < N) {
        uint32_t swap_range = (uint32_t)i + 1, b1 = swap_range + 1;
        uint32_t x = gen.lemire(swap_range * b1);
        uint32_t p0 = x / b1, p1 = x % b1;
        uint32_t a = perm[(size_t)i * stride], b = perm[(size_t)p0 * stride];
        perm[(size_t)i * stride] = b; perm[(size_t)p0 * stride] = a;
        i++;
        a = perm[(size_t)i * stride]; b = perm[(size_t)p1 * stride];
        perm[(size_t)i * stride] = b; perm[(size_t)p1 * stride] = a;
        i++;
    }
}
