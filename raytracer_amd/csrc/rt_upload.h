// rt_upload.h — upload-time re-layout of a reference mesh BVH (Bvh.h:39-49 nodes, children allocated after
// their parent, raytracer.cpp:323-324) into DNode4 records, on the device.
//
// A record covers a reference interior node at even depth and the two levels below it (rt_device.h).  Which
// nodes those are, and where their records go, is computed in parallel:
//   1. k_collapse_parents: every interior node writes itself as the parent of its two children;
//   2. k_collapse_jump (log2(depth) rounds): pointer jumping towards the root, carrying the parity of the
//      distance, until every node points at the root: parity = depth mod 2;
//   3. k_collapse_flag + block scan: record index of node n = number of record roots (interior, even depth) with
//      a smaller node index — records lie in the reference's node order (its depth-first allocation order);
//   4. k_collapse_emit: one thread per record root writes the record: slots L.left, L.right, R.left, R.right in
//      the reference's depth-first order, a leaf child keeps one slot.
// The result depends on nothing but the node array, so it is deterministic; traversal results do not depend on
// where records lie.  Leaves with more than 30 prims (side table) are collected on the host.
#pragma once
#include "rt_device.h"

#define RT_COLLAPSE_BLOCK 1024

__global__ void __launch_bounds__(256) k_collapse_parents(const rt_bvh_node* __restrict__ nodes, uint32_t n, uint32_t* __restrict__ anc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) anc[0] = 0u;                       // the root points at itself, distance 0
    const rt_bvh_node nd = nodes[i];
    if (nd.prim_count == 0) {                      // interior: children at left_or_first, +1 (validated at upload)
        const uint32_t l = (uint32_t)nd.left_or_first;
        anc[l] = i | 0x80000000u;                  // bit 31: parity of the distance to that ancestor
        anc[l + 1] = i | 0x80000000u;
    }
}
__global__ void __launch_bounds__(256) k_collapse_jump(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                       uint32_t* __restrict__ not_done) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = in[i], b = in[a & 0x7FFFFFFFu];
    const uint32_t r = (b & 0x7FFFFFFFu) | ((a ^ b) & 0x80000000u);
    out[i] = r;
    if ((r & 0x7FFFFFFFu) != 0u) *not_done = 1u;   // (plain store of the same value by many threads)
}
// flag = 1 for record roots; exclusive scan inside blocks of RT_COLLAPSE_BLOCK, block totals to sums[]
__global__ void __launch_bounds__(RT_COLLAPSE_BLOCK) k_collapse_flag_scan(const rt_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ anc,
                                                                        uint32_t n, uint32_t* __restrict__ local, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s[RT_COLLAPSE_BLOCK];
    const uint32_t i = blockIdx.x * RT_COLLAPSE_BLOCK + threadIdx.x;
    const uint32_t flag = (i < n && nodes[i].prim_count == 0 && (anc[i] & 0x80000000u) == 0u) ? 1u : 0u;
    s[threadIdx.x] = flag;
    __syncthreads();
    for (uint32_t off = 1; off < RT_COLLAPSE_BLOCK; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    if (i < n) local[i] = (s[threadIdx.x] - flag) | (flag << 31);     // exclusive rank in the block; bit 31 = is a record root
    if (threadIdx.x == RT_COLLAPSE_BLOCK - 1) sums[blockIdx.x] = s[threadIdx.x];
}
// exclusive scan of the block totals (one workgroup, any count); total[0] = number of records
__global__ void __launch_bounds__(RT_COLLAPSE_BLOCK) k_collapse_scan_sums(uint32_t* __restrict__ sums, uint32_t n_blocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t s[RT_COLLAPSE_BLOCK];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += RT_COLLAPSE_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? sums[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < RT_COLLAPSE_BLOCK; off <<= 1) {
            const uint32_t w = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += w;
            __syncthreads();
        }
        if (i < n_blocks) sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += s[RT_COLLAPSE_BLOCK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}
__device__ __forceinline__ uint32_t collapse_leaf_link(const rt_bvh_node& n, uint32_t leaf_base) {
    const uint32_t code = (uint32_t)n.prim_count > 30u ? RT_LEAF_BIG : (uint32_t)n.prim_count;
    return RT_LINK_LEAF | (code << 26) | (leaf_base + (uint32_t)n.left_or_first);
}
__global__ void __launch_bounds__(256) k_collapse_emit(const rt_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ local,
                                                       const uint32_t* __restrict__ sums, uint32_t n, uint32_t leaf_base, uint32_t rec_base,
                                                       DNode4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t li = local[i];
    if (!(li & 0x80000000u)) return;
    const uint32_t rec = (li & 0x7FFFFFFFu) + sums[i / RT_COLLAPSE_BLOCK];
    DNode4 d;
    for (int k = 0; k < 24; k++) d.f[k] = 0.f;
    for (int k = 0; k < 4; k++) d.link[k] = RT_LINK_DONE;
    d.mask = 0;
    d.pad[0] = d.pad[1] = d.pad[2] = 0;
    auto put = [&](int slot, uint32_t ref) {
        const rt_bvh_node c = nodes[ref];
        d.f[slot] = c.bmin[0]; d.f[4 + slot] = c.bmax[0]; d.f[8 + slot] = c.bmin[1];
        d.f[12 + slot] = c.bmax[1]; d.f[16 + slot] = c.bmin[2]; d.f[20 + slot] = c.bmax[2];
        d.mask |= 1u << slot;
        if (c.prim_count > 0) d.link[slot] = collapse_leaf_link(c, leaf_base);
        else d.link[slot] = rec_base + (local[ref] & 0x7FFFFFFFu) + sums[ref / RT_COLLAPSE_BLOCK];   // two levels down: a record root
    };
    const uint32_t L = (uint32_t)nodes[i].left_or_first;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const uint32_t child = L + side;
        const rt_bvh_node c = nodes[child];
        if (c.prim_count > 0) {
            put(2 * side, child);                 // a leaf child keeps one slot; its sibling slot stays empty
        } else {
            put(2 * side, (uint32_t)c.left_or_first);
            put(2 * side + 1, (uint32_t)c.left_or_first + 1);
        }
    }
    out[rec_base + rec] = d;
}
