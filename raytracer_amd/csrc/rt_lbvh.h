// rt_lbvh.h — device side of RT_UPLOAD_REBUILD_MESH_BVH | RT_UPLOAD_REBUILD_ON_DEVICE (include/rt.h): the mesh BVHs
// rebuilt on the GPU as linear BVHs.
//
// NON-PARITY mode, SURVEY §8f-3 (what BuildAllMeshBVHs / SubdivideMesh do on the host, raytracer.cpp:208-339).  Same
// contract as rt_rebuild.h: the trees come out in the REFERENCE's node format (rt_bvh_node: children adjacent and
// after their parent, leaves as ranges of a prim array, left range before right range), so validation, the collapse
// into 4-wide records, the triangle gather, the tie rule by the caller's leaf order and every kernel are the parity
// path's code.  Only the builder differs: Morton codes of the triangle centroids (21 bits per axis), one radix sort,
// Karras' binary radix tree (T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees",
// HPG 2012: every interior node finds its range and split with two binary searches over the sorted keys), boxes
// bottom-up (the second child to arrive merges), subtrees of at most `max_leaf` triangles closed into leaves, and a
// breadth-first numbering that puts every pair of children behind its parent.  A 1 M-triangle mesh: ~2 ms of kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rt.h"
#include "rt_rebuild.h"

namespace rt_lbvh {

struct LBox { float mn[3], mx[3]; };

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// per prim slot p of the BVH: the triangle's box, and the bounds of all centroids (ordered-uint atomics)
__global__ void k_prims(const uint32_t* prim_face, const rt_face* faces, const rt_vertex* vd, uint32_t n, LBox* box, float* cen, uint32_t* cbounds) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const rt_face f = faces[prim_face[p]];
    const float* a = vd[f.i0 - 1].pos;
    const float* b = vd[f.i1 - 1].pos;
    const float* c = vd[f.i2 - 1].pos;
    LBox bx;
    for (int k = 0; k < 3; k++) {
        bx.mn[k] = fminf(fminf(a[k], b[k]), c[k]);
        bx.mx[k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
        const float ck = (a[k] + b[k] + c[k]) * (1.0f / 3.0f);
        cen[(size_t)p * 3 + k] = ck;
        atomicMin(&cbounds[k], f2ord(ck));
        atomicMax(&cbounds[3 + k], f2ord(ck));
    }
    box[p] = bx;
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {     // 21 bits -> every third bit of 63
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void k_keys(const float* cen, const uint32_t* cbounds, uint32_t n, uint64_t* key, uint32_t* val) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint64_t code = 0;
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(cbounds[k]), hi = ord2f(cbounds[3 + k]);
        const float ext = hi - lo;
        float t = ext > 0.0f ? (cen[(size_t)p * 3 + k] - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const uint64_t q = (uint64_t)fminf(t * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - k);
    }
    key[p] = code;
    val[p] = p;
}

// common prefix of sorted positions i and j: Morton code first, the position itself as the tie break (Karras §4)
__device__ __forceinline__ int delta(const uint64_t* key, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = key[i], b = key[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// child encoding: bit 31 set = leaf (sorted position), clear = interior node
__global__ void k_tree(const uint64_t* key, int n, uint32_t* left, uint32_t* right, uint32_t* first, uint32_t* last, uint32_t* parent_of_int, uint32_t* parent_of_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(key, n, i, i + 1) - delta(key, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(key, n, i, i - d);
    int lmax = 2;
    while (delta(key, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(key, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(key, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2, span = l;; ) {
        if (delta(key, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
        span = t;
        t = (span + 1) / 2;
    }
    const int g = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t lc = (lo == g) ? (0x80000000u | (uint32_t)g) : (uint32_t)g;
    const uint32_t rc = (hi == g + 1) ? (0x80000000u | (uint32_t)(g + 1)) : (uint32_t)(g + 1);
    left[i] = lc; right[i] = rc; first[i] = (uint32_t)lo; last[i] = (uint32_t)hi;
    if (lc & 0x80000000u) parent_of_leaf[g] = (uint32_t)i; else parent_of_int[g] = (uint32_t)i;
    if (rc & 0x80000000u) parent_of_leaf[g + 1] = (uint32_t)i; else parent_of_int[g + 1] = (uint32_t)i;
}

// boxes of the interior nodes, bottom-up: the first child to arrive leaves, the second merges and goes on
__global__ void k_boxes(const uint32_t* val, const LBox* box, int n, const uint32_t* left, const uint32_t* right, const uint32_t* parent_of_int,
                        const uint32_t* parent_of_leaf, uint32_t* arrived, LBox* ibox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t node = parent_of_leaf[i];
    for (;;) {
        // The box this thread stored (agent-scope store: written through) has arrived before its arrival is counted, and
        // the sibling's box is read past the caches (agent-scope loads) after the count says it is there.  No fence: a
        // release / acquire fence at agent scope writes back and invalidates the whole L2 (5.5 ms for this kernel).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&arrived[node], 1u) == 0u) return;
        const uint32_t lc = left[node], rc = right[node];
        const LBox* pa = (lc & 0x80000000u) ? &box[val[lc & 0x7FFFFFFFu]] : &ibox[lc];
        const LBox* pb = (rc & 0x80000000u) ? &box[val[rc & 0x7FFFFFFFu]] : &ibox[rc];
        LBox m;
        for (int k = 0; k < 3; k++) {
            const float amn = __hip_atomic_load(&pa->mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), bmn = __hip_atomic_load(&pb->mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float amx = __hip_atomic_load(&pa->mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), bmx = __hip_atomic_load(&pb->mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m.mn[k] = fminf(amn, bmn);
            m.mx[k] = fmaxf(amx, bmx);
        }
        for (int k = 0; k < 3; k++) {
            __hip_atomic_store(&ibox[node].mn[k], m.mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ibox[node].mx[k], m.mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (node == 0) return;
        node = parent_of_int[node];
    }
}

__device__ __forceinline__ void put_node(rt_bvh_node* out, uint32_t o, const LBox& b, int32_t lf, int32_t cnt) {
    rt_bvh_node nd;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = b.mn[k]; nd.bmax[k] = b.mx[k]; }
    nd.left_or_first = lf; nd.prim_count = cnt;
    out[o] = nd;
}
// one level of the breadth-first numbering: (interior node, its slot in the output) -> its pair of children
__global__ void k_emit_level(const uint2* front, uint32_t n_front, uint2* next, uint32_t* counters /* [0]: nodes used, [1]: next frontier */,
                             const uint32_t* val, const LBox* box, const LBox* ibox, const uint32_t* left, const uint32_t* right,
                             const uint32_t* first, const uint32_t* last, uint32_t max_leaf, rt_bvh_node* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_front) return;
    const uint32_t node = front[t].x, o = front[t].y;
    const uint32_t c = atomicAdd(&counters[0], 2u);
    put_node(out, o, ibox[node], (int32_t)c, 0);
    const uint32_t ch[2] = {left[node], right[node]};
    for (uint32_t k = 0; k < 2; k++) {
        if (ch[k] & 0x80000000u) {
            const uint32_t pos = ch[k] & 0x7FFFFFFFu;
            put_node(out, c + k, box[val[pos]], (int32_t)pos, 1);
        } else {
            const uint32_t f = first[ch[k]], sz = last[ch[k]] - f + 1u;
            if (sz <= max_leaf) put_node(out, c + k, ibox[ch[k]], (int32_t)f, (int32_t)sz);
            else next[atomicAdd(&counters[1], 1u)] = make_uint2(ch[k], c + k);
        }
    }
}
__global__ void k_root_leaf(const uint32_t* val, const LBox* box, const LBox* ibox, uint32_t n, rt_bvh_node* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (n == 1) put_node(out, 0, box[val[0]], 0, 1);
    else put_node(out, 0, ibox[0], 0, (int32_t)n);
}
__global__ void k_sorted_faces(const uint32_t* val, const uint32_t* prim_face, uint32_t n, uint32_t* sorted_face) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sorted_face[i] = prim_face[val[i]];
}

struct Scratch {
    void* p = nullptr;
    size_t bytes = 0;
    ~Scratch() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
};

struct Lap {
    bool on = getenv("RT_UPLOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[rt upload]   lbvh %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

#define RT_LBVH_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = hipGetErrorString(e_); return false; } } while (0)

// All mesh BVHs of the scene, on the current device; the result in the same form as rt_rebuild::rebuild.
inline bool rebuild(const rt_scene_desc* s, uint32_t max_leaf, rt_rebuild::MeshLevel& out, const char*& err) {
    out.bvhs.assign(s->mesh_bvhs, s->mesh_bvhs + s->n_mesh_bvhs);
    out.prim_face.assign(s->mesh_prim_face, s->mesh_prim_face + s->n_mesh_prims);
    out.rank.assign(s->n_mesh_prims, 0);
    out.nodes.clear();
    uint32_t n_max = 0;
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) n_max = std::max(n_max, s->mesh_bvhs[b].prim_count);
    if (n_max == 0) { out.nodes.assign(s->mesh_nodes, s->mesh_nodes + s->n_mesh_nodes); return true; }

    Lap lap;
    Scratch d_faces, d_vd, d_work, d_sort;
    RT_LBVH_CHK(d_faces.ensure((size_t)s->n_faces * sizeof(rt_face)));
    RT_LBVH_CHK(d_vd.ensure((size_t)s->n_vertices * sizeof(rt_vertex)));
    RT_LBVH_CHK(hipMemcpy(d_faces.p, s->faces, (size_t)s->n_faces * sizeof(rt_face), hipMemcpyHostToDevice));
    RT_LBVH_CHK(hipMemcpy(d_vd.p, s->vertex_data, (size_t)s->n_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice));
    // one allocation, carved: per prim / per interior node arrays
    const size_t N = n_max;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pf = carve(N * 4), o_box = carve(N * sizeof(LBox)), o_cen = carve(N * 12), o_cb = carve(32), o_key = carve(N * 8), o_key2 = carve(N * 8),
                 o_val = carve(N * 4), o_val2 = carve(N * 4), o_left = carve(N * 4), o_right = carve(N * 4), o_first = carve(N * 4), o_last = carve(N * 4),
                 o_pint = carve(N * 4), o_pleaf = carve(N * 4), o_arr = carve(N * 4), o_ibox = carve(N * sizeof(LBox)), o_fa = carve(N * 8), o_fb = carve(N * 8),
                 o_cnt = carve(32), o_out = carve(2 * N * sizeof(rt_bvh_node)), o_sf = carve(N * 4);
    RT_LBVH_CHK(d_work.ensure(off));
    char* W = (char*)d_work.p;
    size_t sort_bytes = 0;
    RT_LBVH_CHK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)N, 0, 63));
    RT_LBVH_CHK(d_sort.ensure(sort_bytes));
    lap("alloc + faces/vertices up");

    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
        rt_mesh_bvh& mb = out.bvhs[b];
        const size_t node_off = out.nodes.size();
        if (mb.prim_count == 0) {                  // an empty mesh keeps whatever nodes the caller gave it
            out.nodes.insert(out.nodes.end(), s->mesh_nodes + mb.node_offset, s->mesh_nodes + mb.node_offset + mb.node_count);
            mb.node_offset = (uint32_t)node_off;
            continue;
        }
        const uint32_t n = mb.prim_count;
        const int grid = (int)((n + 255) / 256);
        uint32_t* pf = (uint32_t*)(W + o_pf);
        LBox* box = (LBox*)(W + o_box);
        float* cen = (float*)(W + o_cen);
        uint32_t* cb = (uint32_t*)(W + o_cb);
        uint64_t *key = (uint64_t*)(W + o_key), *key2 = (uint64_t*)(W + o_key2);
        uint32_t *val = (uint32_t*)(W + o_val), *val2 = (uint32_t*)(W + o_val2);
        uint32_t *left = (uint32_t*)(W + o_left), *right = (uint32_t*)(W + o_right), *first = (uint32_t*)(W + o_first), *last = (uint32_t*)(W + o_last);
        uint32_t *pint = (uint32_t*)(W + o_pint), *pleaf = (uint32_t*)(W + o_pleaf), *arrived = (uint32_t*)(W + o_arr);
        LBox* ibox = (LBox*)(W + o_ibox);
        uint2 *fa = (uint2*)(W + o_fa), *fb = (uint2*)(W + o_fb);
        uint32_t* cnt = (uint32_t*)(W + o_cnt);
        rt_bvh_node* d_out = (rt_bvh_node*)(W + o_out);
        uint32_t* sf = (uint32_t*)(W + o_sf);

        RT_LBVH_CHK(hipMemcpy(pf, s->mesh_prim_face + mb.prim_offset, (size_t)n * 4, hipMemcpyHostToDevice));
        const uint32_t cb0[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
        RT_LBVH_CHK(hipMemcpy(cb, cb0, 32, hipMemcpyHostToDevice));
        k_prims<<<grid, 256>>>(pf, (const rt_face*)d_faces.p + mb.face_offset, (const rt_vertex*)d_vd.p, n, box, cen, cb);
        k_keys<<<grid, 256>>>(cen, cb, n, key, val);
        RT_LBVH_CHK(hipcub::DeviceRadixSort::SortPairs(d_sort.p, sort_bytes, key, key2, val, val2, (int)n, 0, 63));
        lap("boxes, keys, sort");
        uint32_t n_nodes = 1;
        if (n <= max_leaf || n == 1) {
            // (boxes of a tree that small: one interior pass is not worth it — the root leaf takes the union on the host below)
            if (n > 1) {
                k_tree<<<grid, 256>>>(key2, (int)n, left, right, first, last, pint, pleaf);
                RT_LBVH_CHK(hipMemset(arrived, 0, (size_t)n * 4));
                k_boxes<<<grid, 256>>>(val2, box, (int)n, left, right, pint, pleaf, arrived, ibox);
            }
            k_root_leaf<<<1, 64>>>(val2, box, ibox, n, d_out);
        } else {
            k_tree<<<grid, 256>>>(key2, (int)n, left, right, first, last, pint, pleaf);
            RT_LBVH_CHK(hipMemset(arrived, 0, (size_t)n * 4));
            k_boxes<<<grid, 256>>>(val2, box, (int)n, left, right, pint, pleaf, arrived, ibox);
            lap("radix tree + boxes");
            const uint2 root = make_uint2(0u, 0u);
            RT_LBVH_CHK(hipMemcpy(fa, &root, 8, hipMemcpyHostToDevice));
            uint32_t h_cnt[2] = {1u, 0u};
            RT_LBVH_CHK(hipMemcpy(cnt, h_cnt, 8, hipMemcpyHostToDevice));
            uint32_t n_front = 1;
            while (n_front != 0) {
                k_emit_level<<<(int)((n_front + 255) / 256), 256>>>(fa, n_front, fb, cnt, val2, box, ibox, left, right, first, last, max_leaf, d_out);
                RT_LBVH_CHK(hipMemcpy(h_cnt, cnt, 8, hipMemcpyDeviceToHost));
                n_front = h_cnt[1];
                h_cnt[1] = 0;
                RT_LBVH_CHK(hipMemcpy(cnt + 1, &h_cnt[1], 4, hipMemcpyHostToDevice));
                std::swap(fa, fb);
            }
            n_nodes = h_cnt[0];
        }
        lap("breadth-first emit");
        k_sorted_faces<<<grid, 256>>>(val2, pf, n, sf);
        RT_LBVH_CHK(hipGetLastError());
        out.nodes.resize(node_off + n_nodes);
        RT_LBVH_CHK(hipMemcpy(out.nodes.data() + node_off, d_out, (size_t)n_nodes * sizeof(rt_bvh_node), hipMemcpyDeviceToHost));
        // leaf order of the new tree, and each slot's position in the caller's leaf order — which is what the sorted values are
        RT_LBVH_CHK(hipMemcpy(out.prim_face.data() + mb.prim_offset, sf, (size_t)n * 4, hipMemcpyDeviceToHost));
        uint32_t* rk = out.rank.data() + mb.prim_offset;
        RT_LBVH_CHK(hipMemcpy(rk, val2, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (mb.prim_offset != 0) for (uint32_t p = 0; p < n; p++) rk[p] += mb.prim_offset;
        mb.node_offset = (uint32_t)node_off;
        mb.node_count = n_nodes;
        lap("download + rank");
    }
    return true;
}
#undef RT_LBVH_CHK

}  // namespace rt_lbvh
