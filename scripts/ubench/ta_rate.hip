// ta_rate.hip — microbenchmark: how fast does a CU's texture-addresser / vector L1 serve
// address-divergent loads of BVH-node-shaped records?  (DESIGN.md §6 uses the result.)
//   A  each lane reads its own 128-B record with 8 dwordx4 loads   (what traverse() does per node step)
//   B  8 lanes share one record per load (16 B each, 128 B contiguous), 8 loads cover 64 records
//   C  each lane reads its own 36-B record with 9 dword loads      (triangle, as compiled from float[9])
//   D  same 36-B record as dwordx4 + dwordx4 + dword
//   E  each lane reads its own 48-B record with 3 dwordx4 loads    (padded triangle)
//   F  each lane reads its own 128-B record from LDS with 8 ds_read_b128
// Footprints: 16 KB (L1), 2 MB (L2), 256 MB (MALL/HBM).  Prints ns per wave-level record fetch per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const uint32_t* __restrict__ buf, uint32_t nrec_mask, int iters, uint32_t* out) {
    __shared__ uint4 lds[2048];   // 32 KB
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t accu = 0;
    if (MODE == 5) { for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(i, i, i, i); __syncthreads(); }
    for (int it = 0; it < iters; it++) {
        uint32_t r = lcg(s) & nrec_mask;
        if (MODE == 0) {
            const uint4* p = reinterpret_cast<const uint4*>(buf + (size_t)r * 32);
            uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6], a7 = p[7];
            accu += a0.x ^ a1.y ^ a2.z ^ a3.w ^ a4.x ^ a5.y ^ a6.z ^ a7.w;
        } else if (MODE == 1) {
            // round j: lane serves record of lane (8j + lane/8), chunk lane%8
            #pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t rr = __shfl(r, 8 * j + (lane >> 3));
                const uint4* p = reinterpret_cast<const uint4*>(buf + (size_t)rr * 32);
                uint4 a = p[lane & 7];
                accu += a.x ^ a.w;
            }
        } else if (MODE == 2) {
            const uint32_t* p = buf + (size_t)r * 9;
            #pragma unroll
            for (int j = 0; j < 9; j++) accu ^= __builtin_nontemporal_load(p + j) * (j + 1);
        } else if (MODE == 3) {
            const uint32_t* p = buf + (size_t)r * 9;
            uint4 a, b; uint32_t c;
            asm volatile("global_load_dwordx4 %0, %3, off\n global_load_dwordx4 %1, %3, off offset:16\n global_load_dword %2, %3, off offset:32\n s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
            accu += a.x ^ a.w ^ b.y ^ b.z ^ c;
        } else if (MODE == 4) {
            const uint4* p = reinterpret_cast<const uint4*>(buf + (size_t)r * 12);
            uint4 a0 = p[0], a1 = p[1], a2 = p[2];
            accu += a0.x ^ a1.y ^ a2.z;
        } else if (MODE == 5) {
            const uint4* p = &lds[(r & 255) * 8];
            uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6], a7 = p[7];
            accu += a0.x ^ a1.y ^ a2.z ^ a3.w ^ a4.x ^ a5.y ^ a6.z ^ a7.w;
        }
    }
    if (accu == 0x12345678u) out[0] = accu;
}

template <int MODE>
static void run(const char* name, const uint32_t* buf, size_t rec_words, uint32_t* out) {
    const size_t foots[3] = {16u << 10, 2u << 20, 256u << 20};
    for (int f = 0; f < 3; f++) {
        uint32_t nrec = 1; while ((size_t)nrec * 2 * rec_words * 4 <= foots[f]) nrec *= 2;
        const int iters = 2000, blocks = 256 * 4;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k<MODE><<<blocks, 256>>>(buf, nrec - 1, 100, out);
        CK(hipEventRecord(e0));
        k<MODE><<<blocks, 256>>>(buf, nrec - 1, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // per CU: 16 waves x iters wave-level record fetches
        double ns_per_fetch = ms * 1e6 / (16.0 * iters);
        printf("%-44s footprint %8zu KB  %8.3f ms  %7.1f ns per wave-fetch per CU  (%.1f GB/s useful)\n", name, (size_t)nrec * rec_words * 4 >> 10, ms,
               ns_per_fetch, (double)blocks * 256 * iters * rec_words * 4 / (ms * 1e6));
    }
}

int main() {
    uint32_t *buf, *out;
    CK(hipMalloc(&buf, (512u << 20)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, 512u << 20));
    run<0>("A lane-private 128B, 8 x dwordx4", buf, 32, out);
    run<1>("B 8 lanes per 128B record, 8 x dwordx4", buf, 32, out);
    run<2>("C lane-private 36B, 9 x dword", buf, 9, out);
    run<3>("D lane-private 36B, x4 + x4 + x1", buf, 9, out);
    run<4>("E lane-private 48B, 3 x dwordx4", buf, 12, out);
    run<5>("F lane-private 128B from LDS, 8 x ds_read_b128", buf, 32, out);
    return 0;
}
