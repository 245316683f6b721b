// Known answers of the libstdc++ RNG adaptors the reference's pixel loop uses (raytracer.cpp:44-69):
//   std::mt19937 rng(seed); std::shuffle(v.begin(), v.end(), rng) over [0..N);
//   std::uniform_real_distribution<float> dist(0,1); dist(rng) x 5
// Written by libstdc++ itself (g++ 11.4, the compiler the reference is built with here), not by this
// repository's restatement:   g++ -std=c++11 -O2 tests/golden/make_rng_kat.cpp -o /tmp/kat && /tmp/kat > tests/golden/rng_kat.json
// Each row: seed, N, the whole shuffled permutation, then the bit patterns of the next five floats.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main() {
    const int Ns[] = {1, 2, 3, 4, 5, 9, 16, 25, 36, 64, 100, 900};
    const uint32_t seeds[] = {0u, 1u, 799u * 800u + 799u, 123456789u, 4294967295u};
    std::printf("{\"rows\": [\n");
    bool first = true;
    for (int N : Ns)
        for (uint32_t seed : seeds) {
            std::mt19937 rng(seed);
            std::uniform_real_distribution<float> dist(0.0f, 1.0f);
            std::vector<int> v(N);
            for (int i = 0; i < N; i++) v[i] = i;
            std::shuffle(v.begin(), v.end(), rng);
            std::printf("%s{\"seed\": %u, \"n\": %d, \"perm\": [", first ? "" : ",\n", seed, N);
            for (int i = 0; i < N; i++) std::printf("%s%d", i ? "," : "", v[i]);
            std::printf("], \"float_bits\": [");
            for (int i = 0; i < 5; i++) {
                float f = dist(rng);
                uint32_t b;
                std::memcpy(&b, &f, 4);
                std::printf("%s%u", i ? "," : "", b);
            }
            std::printf("]}");
            first = false;
        }
    std::printf("\n]}\n");
    return 0;
}
