// rt_dsah.h — device side of RT_UPLOAD_REBUILD_MESH_BVH | RT_UPLOAD_REBUILD_ON_DEVICE (include/rt.h): the mesh BVHs
// rebuilt on the GPU with the binned surface-area heuristic of rt_rebuild.h.
//
// NON-PARITY mode, SURVEY §8f-3 (what BuildAllMeshBVHs / SubdivideMesh do on the host, raytracer.cpp:208-339).  Same
// contract as rt_rebuild.h and rt_lbvh.h: the trees come out in the REFERENCE's node format (rt_bvh_node: children
// adjacent and after their parent, leaves as ranges of a prim array, left range before right range), so validation, the
// collapse into 4-wide records, the triangle gather, the tie rule by the caller's leaf order and every kernel are the
// parity path's code.  The split rule is rt_rebuild.h's, decision for decision (16 bins x 3 axes over the centroid
// bounds, cost = area x count of both sides, bins scanned from the top, the first axis wins ties, a leaf of at most
// `max_leaf` triangles when no split is cheaper, coincident centroids halved by position), and every quantity it uses
// is a minimum, a maximum or a count — independent of the order in which a GPU adds things up — so the tree is the
// host builder's tree; only the numbering of the nodes and the order of the triangles inside a leaf differ (ties
// between equal t are settled by the caller's leaf order, `rank`, as for every rebuilt tree).
//
// How it runs.  Top-down, in two regimes:
//  * nodes of more than 64 triangles, level by level over the whole mesh: one thread per triangle adds its centroid to
//    its node's centroid bounds, then its box to the node's 3 x 16 bins (atomics on order-preserving integers; a
//    workgroup whose 256 triangles lie in one node bins in LDS first); one thread per node scans the bins and
//    decides; a prefix sum over the nodes numbers the children (breadth-first, deterministic); a prefix sum over the
//    triangles partitions every node's range at once (stable).
//  * subtrees of at most 64 triangles: one wave each, a triangle per lane (box, centroid and id stay in registers and
//    move between lanes when a node is partitioned), bins in LDS, the three axes scanned by three lanes, nodes written
//    to a private strip and then copied behind the upper levels with one more prefix sum (deterministic again).
// A 1 M-triangle mesh: ~20 levels; see profiles/r03 for the time.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <vector>
#include "rt.h"
#include "rt_lbvh.h"      // LBox, f2ord / ord2f, k_sorted_faces, Scratch, Lap
#include "rt_rebuild.h"

namespace rt_dsah {

using rt_lbvh::LBox;
using rt_lbvh::f2ord;
using rt_lbvh::ord2f;

constexpr int BINS = rt_rebuild::BINS;      // 16
constexpr uint32_t SMALL = 64;              // subtrees of at most this many triangles are built by one wave
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t ORD_PINF = 0xFF800000u;  // f2ord(+inf)
constexpr uint32_t ORD_NINF = 0x007FFFFFu;  // f2ord(-inf)

struct Bin { uint32_t mn[3], mx[3], cnt; };             // box as order-preserving integers (atomicMin / atomicMax)
struct Prep { float cmn[3], scale[3]; };                // per frontier node: centroid bounds -> bin scale (rt_rebuild.h:116-122)
enum : uint32_t { SPLIT_X = 0, SPLIT_Y = 1, SPLIT_Z = 2, SPLIT_HALVE = 3, SPLIT_LEAF = 4 };
struct Split { uint32_t code, split, first, count, left_count, child; LBox l, r; };

__device__ __forceinline__ float half_area(const LBox& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void box_reset(LBox& b) { for (int a = 0; a < 3; a++) { b.mn[a] = __builtin_inff(); b.mx[a] = -__builtin_inff(); } }
__device__ __forceinline__ void box_add(LBox& b, const LBox& o) { for (int a = 0; a < 3; a++) { b.mn[a] = fminf(b.mn[a], o.mn[a]); b.mx[a] = fmaxf(b.mx[a], o.mx[a]); } }
__device__ __forceinline__ LBox bin_box(const Bin& n) {
    LBox b;
    for (int a = 0; a < 3; a++) { b.mn[a] = ord2f(n.mn[a]); b.mx[a] = ord2f(n.mx[a]); }
    return b;
}
__device__ __forceinline__ int bin_of(float c, float lo, float scale) {       // rt_rebuild.h: Builder::bin_of
    const int k = (int)((c - lo) * scale);
    return k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
}
__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o != 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o != 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// One axis of the split search (rt_rebuild.h:141-158): the best cost over the 15 places between the bins, scanned
// from the top, strict improvement only.  `bins` = the 16 bins of that axis.
__device__ inline void axis_sweep(const Bin* bins, float& best_cost, int& best_split) {
    float la[BINS - 1];
    uint32_t lc[BINS - 1];
    LBox acc; box_reset(acc);
    uint32_t cnt = 0;
    for (int k = 0; k < BINS - 1; k++) {
        const Bin b = bins[k];
        if (b.cnt) box_add(acc, bin_box(b));
        cnt += b.cnt;
        la[k] = cnt ? half_area(acc) : 0.0f;
        lc[k] = cnt;
    }
    box_reset(acc); cnt = 0;
    for (int k = BINS - 1; k > 0; k--) {
        const Bin b = bins[k];
        if (b.cnt) box_add(acc, bin_box(b));
        cnt += b.cnt;
        if (cnt == 0 || lc[k - 1] == 0) continue;
        const float cost = la[k - 1] * (float)lc[k - 1] + half_area(acc) * (float)cnt;
        if (cost < best_cost) { best_cost = cost; best_split = k; }
    }
}
__device__ inline void union_bins(const Bin* bins, int lo, int hi, LBox& out, uint32_t& count) {
    box_reset(out); count = 0;
    for (int k = lo; k < hi; k++) { const Bin b = bins[k]; if (b.cnt) { box_add(out, bin_box(b)); count += b.cnt; } }
}
// leaf instead of a split when no partition is cheaper than testing the triangles (rt_rebuild.h:167-173)
__device__ __forceinline__ bool prefers_leaf(uint32_t count, uint32_t max_leaf, float ci, float best_cost, const LBox& node_box) {
    if (count > max_leaf) return false;
    const float pa = half_area(node_box);
    return !(pa > 0.0f) || ci * best_cost / pa + 1.0f >= ci * (float)count;
}

// ---------------------------------------------------------------------------------------------- set-up
// Same-address atomics are what these passes would spend their time on (~60 ns each once they queue on one line: 16 000 waves
// adding to the root box took 1 ms), so every wave / workgroup walks a CHUNK of consecutive positions, keeps what it adds
// up for the node its positions lie in, and goes to memory only when that node changes or the chunk ends.
constexpr uint32_t CHUNK = 16;              // positions per lane (steps of 64 per wave) in k_prims / k_cbounds / k_bin

// per prim slot p of the BVH: the triangle's box and centroid (rt_rebuild.h:208-215); the root box
__global__ void __launch_bounds__(256) k_prims(const uint32_t* prim_face, const rt_face* faces, const rt_vertex* vd, uint32_t n, LBox* box, float* cen,
                                               uint32_t* root_box /* [6] ordered */, uint32_t* order, uint32_t* pnode) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    LBox acc; box_reset(acc);
    for (uint32_t k = 0; k < CHUNK; k++) {
        const uint32_t p = (wave * CHUNK + k) * 64u + lane;
        if (p >= n) break;
        const rt_face f = faces[prim_face[p]];
        const float* a = vd[f.i0 - 1].pos;
        const float* b = vd[f.i1 - 1].pos;
        const float* c = vd[f.i2 - 1].pos;
        LBox bx;
        for (int d = 0; d < 3; d++) {
            bx.mn[d] = fminf(fminf(a[d], b[d]), c[d]);
            bx.mx[d] = fmaxf(fmaxf(a[d], b[d]), c[d]);
            cen[(size_t)p * 3 + d] = (a[d] + b[d] + c[d]) * (1.0f / 3.0f);
        }
        box[p] = bx;
        box_add(acc, bx);
        order[p] = p;
        pnode[p] = n > SMALL ? 0u : NONE;
    }
    for (int d = 0; d < 3; d++) {
        const float mn = wave_min(acc.mn[d]), mx = wave_max(acc.mx[d]);
        if (lane == 0u && mn <= mx) { atomicMin(&root_box[d], f2ord(mn)); atomicMax(&root_box[3 + d], f2ord(mx)); }
    }
}
__global__ void k_root(const uint32_t* root_box, uint32_t n, rt_bvh_node* out, uint32_t* ncb, uint32_t* nslot, uint32_t* frontier, uint32_t* small_list, uint32_t* counters) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    rt_bvh_node nd;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = ord2f(root_box[k]); nd.bmax[k] = ord2f(root_box[3 + k]); }
    nd.left_or_first = 0; nd.prim_count = (int32_t)n;
    out[0] = nd;
    for (int k = 0; k < 3; k++) { ncb[k] = ORD_PINF; ncb[3 + k] = ORD_NINF; }
    nslot[0] = 0;
    frontier[0] = 0;
    small_list[0] = 0;
    counters[0] = 1;                               // nodes used
    counters[1] = n > SMALL ? 1u : 0u;             // frontier size
    counters[2] = (n > 2 && n <= SMALL) ? 1u : 0u; // small subtrees
}

// ---------------------------------------------------------------------------------------------- upper levels
// centroid bounds of every frontier node (rt_rebuild.h:107-115)
__device__ __forceinline__ void cb_flush(uint32_t node, const float* mn, const float* mx, uint32_t lane, uint32_t* ncb) {
    if (node == NONE) return;
    for (int d = 0; d < 3; d++) {
        const float lo = wave_min(mn[d]), hi = wave_max(mx[d]);
        if (lane == 0u && lo <= hi) { atomicMin(&ncb[(size_t)node * 6 + d], f2ord(lo)); atomicMax(&ncb[(size_t)node * 6 + 3 + d], f2ord(hi)); }
    }
}
__global__ void __launch_bounds__(256) k_cbounds(const uint32_t* order, const uint32_t* pnode, const float* cen, uint32_t n, uint32_t* ncb) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    uint32_t acc_node = NONE;                                      // wave-uniform
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t k = 0; k < CHUNK; k++) {
        const uint32_t p0 = (wave * CHUNK + k) * 64u;
        if (p0 >= n) break;
        const uint32_t p = p0 + lane;
        const uint32_t node = p < n ? pnode[p] : NONE;
        float c[3] = {0.f, 0.f, 0.f};
        if (node != NONE) { const float* q = &cen[(size_t)order[p] * 3]; c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; }
        const uint32_t node0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
        if (__all(node == node0)) {              // these 64 positions lie in one node (or in none)
            if (node0 != acc_node) {
                cb_flush(acc_node, mn, mx, lane, ncb);
                acc_node = node0;
                for (int d = 0; d < 3; d++) { mn[d] = __builtin_inff(); mx[d] = -__builtin_inff(); }
            }
            if (node0 != NONE) for (int d = 0; d < 3; d++) { mn[d] = fminf(mn[d], c[d]); mx[d] = fmaxf(mx[d], c[d]); }
        } else if (node != NONE) {
            for (int d = 0; d < 3; d++) { atomicMin(&ncb[(size_t)node * 6 + d], f2ord(c[d])); atomicMax(&ncb[(size_t)node * 6 + 3 + d], f2ord(c[d])); }
        }
    }
    cb_flush(acc_node, mn, mx, lane, ncb);
}
// per frontier slot: bin scale from the centroid bounds; its 48 bins emptied (one thread per bin)
__global__ void k_prep(const uint32_t* frontier, uint32_t nf, const uint32_t* ncb, Prep* prep, Bin* bins) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = t / (3 * BINS), b = t % (3 * BINS);
    if (slot >= nf) return;
    Bin e;
    for (int k = 0; k < 3; k++) { e.mn[k] = ORD_PINF; e.mx[k] = ORD_NINF; }
    e.cnt = 0;
    bins[(size_t)slot * 3 * BINS + b] = e;
    if (b == 0) {
        const uint32_t node = frontier[slot];
        Prep pr;
        for (int a = 0; a < 3; a++) {
            const float lo = ord2f(ncb[(size_t)node * 6 + a]), hi = ord2f(ncb[(size_t)node * 6 + 3 + a]);
            const float ext = hi - lo;
            pr.cmn[a] = lo;
            pr.scale[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
        }
        prep[slot] = pr;
    }
}
__device__ __forceinline__ void bin_add(Bin* b, const LBox& bx) {
    for (int k = 0; k < 3; k++) { atomicMin(&b->mn[k], f2ord(bx.mn[k])); atomicMax(&b->mx[k], f2ord(bx.mx[k])); }
    atomicAdd(&b->cnt, 1u);
}
// every triangle of a frontier node into that node's bins, all three axes in one pass (rt_rebuild.h:123-135).  A WAVE walks
// CHUNK x 64 consecutive positions with bins of its own in LDS: while the 64 positions of a step lie in one node it bins
// there and flushes to the node's bins in memory when the node changes; a step that straddles nodes goes to memory directly.
#define RT_DSAH_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
__global__ void __launch_bounds__(256) k_bin(const uint32_t* order, const uint32_t* pnode, const uint32_t* nslot, const float* cen, const LBox* box,
                                             const Prep* prep, uint32_t n, Bin* bins) {
    __shared__ Bin s_bins[4][3 * BINS];
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    Bin* lbins = s_bins[threadIdx.x >> 6];
    uint32_t acc_node = NONE;                                      // wave-uniform
    auto reset = [&]() {
        if (lane < 3 * BINS) {
            Bin e;
            for (int k = 0; k < 3; k++) { e.mn[k] = ORD_PINF; e.mx[k] = ORD_NINF; }
            e.cnt = 0;
            lbins[lane] = e;
        }
    };
    auto flush = [&]() {
        if (acc_node != NONE && lane < 3 * BINS) {
            const Bin e = lbins[lane];
            if (e.cnt) {
                Bin* g = bins + (size_t)nslot[acc_node] * 3 * BINS + lane;
                for (int k = 0; k < 3; k++) { atomicMin(&g->mn[k], e.mn[k]); atomicMax(&g->mx[k], e.mx[k]); }
                atomicAdd(&g->cnt, e.cnt);
            }
        }
    };
    for (uint32_t k = 0; k < CHUNK; k++) {
        const uint32_t p0 = (wave * CHUNK + k) * 64u;
        if (p0 >= n) break;
        const uint32_t p = p0 + lane;
        const uint32_t node = p < n ? pnode[p] : NONE;
        const uint32_t node0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
        const bool uniform = __all(p >= n || node == node0);
        if (uniform && node0 != acc_node) {
            RT_DSAH_LDS_SYNC();
            flush();
            RT_DSAH_LDS_SYNC();
            reset();
            acc_node = node0;
            RT_DSAH_LDS_SYNC();
        }
        if (node != NONE) {
            const uint32_t slot = nslot[node];
            const Prep pr = prep[slot];
            const uint32_t id = order[p];
            const LBox bx = box[id];
            for (int a = 0; a < 3; a++) {
                if (pr.scale[a] == 0.0f) continue;
                const int b = a * BINS + bin_of(cen[(size_t)id * 3 + a], pr.cmn[a], pr.scale[a]);
                if (uniform) bin_add(&lbins[b], bx); else bin_add(bins + (size_t)slot * 3 * BINS + b, bx);
            }
        }
    }
    RT_DSAH_LDS_SYNC();
    flush();
}
// one thread per frontier node: the split decision (rt_rebuild.h:139-180).  packed[slot] = splits | big children << 21 |
// small children << 42, for the prefix sum that numbers the children and fills the next frontier.
__global__ void k_split(const uint32_t* frontier, uint32_t nf, const rt_bvh_node* out, const Bin* bins, const Prep* prep, const uint32_t* order,
                        const LBox* box, uint32_t max_leaf, float ci, Split* split, unsigned long long* packed) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nf) return;
    const uint32_t node = frontier[slot];
    const rt_bvh_node nd = out[node];
    const uint32_t first = (uint32_t)nd.left_or_first, count = (uint32_t)nd.prim_count;
    const Bin* B = bins + (size_t)slot * 3 * BINS;
    const Prep pr = prep[slot];
    float best_cost = __builtin_inff();
    int best_axis = -1, best_split = 0;
    for (int a = 0; a < 3; a++) {
        if (pr.scale[a] == 0.0f) continue;
        float c = best_cost; int s = 0;
        axis_sweep(B + a * BINS, c, s);
        if (c < best_cost) { best_cost = c; best_axis = a; best_split = s; }
    }
    Split sp;
    sp.first = first; sp.count = count; sp.child = 0; sp.split = (uint32_t)best_split;
    box_reset(sp.l); box_reset(sp.r);
    LBox nb;
    for (int k = 0; k < 3; k++) { nb.mn[k] = nd.bmin[k]; nb.mx[k] = nd.bmax[k]; }
    if (best_axis < 0) {
        // all centroids coincide (raytracer.cpp:317 stops here with one big leaf): halve by position
        if (count <= max_leaf) { sp.code = SPLIT_LEAF; sp.left_count = count; }
        else {
            sp.code = SPLIT_HALVE; sp.left_count = count / 2;
            for (uint32_t i = 0; i < count; i++) box_add(i < sp.left_count ? sp.l : sp.r, box[order[first + i]]);
        }
    } else if (prefers_leaf(count, max_leaf, ci, best_cost, nb)) {
        sp.code = SPLIT_LEAF; sp.left_count = count;
    } else {
        sp.code = (uint32_t)best_axis;
        uint32_t lc, rc;
        union_bins(B + best_axis * BINS, 0, best_split, sp.l, lc);
        union_bins(B + best_axis * BINS, best_split, BINS, sp.r, rc);
        sp.left_count = lc;
    }
    split[slot] = sp;
    unsigned long long pk = 0;
    if (sp.code != SPLIT_LEAF) {
        const uint32_t lc = sp.left_count, rc = count - lc;
        pk = 1ull + ((unsigned long long)((lc > SMALL) + (rc > SMALL)) << 21) + ((unsigned long long)((lc > 2 && lc <= SMALL) + (rc > 2 && rc <= SMALL)) << 42);
    }
    packed[slot] = pk;
}
using rt_lbvh::put_node;
__global__ void k_totals(const unsigned long long* packed, const unsigned long long* scan, uint32_t nf, unsigned long long* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = packed[nf - 1] + scan[nf - 1];
}
// children of the frontier nodes that split: numbered by the prefix sum (pairs behind everything made so far, in
// frontier order), large ones into the next frontier, small ones onto the list of wave-built subtrees
__global__ void k_children(const uint32_t* frontier, uint32_t nf, const unsigned long long* scan, Split* split, uint32_t nodes_used, uint32_t n_small,
                           rt_bvh_node* out, uint32_t* ncb, uint32_t* nslot, uint32_t* next_frontier, uint32_t* small_list) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nf) return;
    Split sp = split[slot];
    if (sp.code == SPLIT_LEAF) return;
    const unsigned long long sc = scan[slot];
    const uint32_t c = nodes_used + 2u * (uint32_t)(sc & 0x1FFFFFull);
    uint32_t big_i = (uint32_t)((sc >> 21) & 0x1FFFFFull), small_i = n_small + (uint32_t)((sc >> 42) & 0x1FFFFFull);
    const uint32_t node = frontier[slot];
    out[node].left_or_first = (int32_t)c;
    out[node].prim_count = 0;
    const uint32_t cnt[2] = {sp.left_count, sp.count - sp.left_count};
    const uint32_t fst[2] = {sp.first, sp.first + sp.left_count};
    for (uint32_t k = 0; k < 2; k++) {
        put_node(out, c + k, k == 0 ? sp.l : sp.r, (int32_t)fst[k], (int32_t)cnt[k]);
        if (cnt[k] > SMALL) {
            next_frontier[big_i] = c + k;
            nslot[c + k] = big_i++;
            for (int a = 0; a < 3; a++) { ncb[(size_t)(c + k) * 6 + a] = ORD_PINF; ncb[(size_t)(c + k) * 6 + 3 + a] = ORD_NINF; }
        } else if (cnt[k] > 2) {
            small_list[small_i++] = c + k;
        }
    }
    split[slot].child = c;
}
// 1 for a triangle that goes to the left child of its node
__global__ void k_flags(const uint32_t* order, const uint32_t* pnode, const uint32_t* nslot, const float* cen, const Prep* prep, const Split* split,
                        uint32_t n, uint32_t* flag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t node = pnode[p];
    uint32_t f = 0;
    if (node != NONE) {
        const uint32_t slot = nslot[node];
        const uint32_t code = split[slot].code;
        if (code <= SPLIT_Z) f = bin_of(cen[(size_t)order[p] * 3 + code], prep[slot].cmn[code], prep[slot].scale[code]) < (int)split[slot].split ? 1u : 0u;
        else if (code == SPLIT_HALVE) f = (p - split[slot].first) < split[slot].left_count ? 1u : 0u;
    }
    flag[p] = f;
}
// stable partition of every frontier node's range at once (rt_rebuild.h:174-179 swaps in place; only the sets matter)
__global__ void k_scatter(const uint32_t* order, const uint32_t* pnode, const uint32_t* nslot, const Split* split, const uint32_t* flag,
                          const uint32_t* scan, uint32_t n, uint32_t* order2, uint32_t* pnode2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t node = pnode[p];
    uint32_t q = p, pn = NONE;
    if (node != NONE) {
        const Split& sp = split[nslot[node]];
        if (sp.code != SPLIT_LEAF) {
            const uint32_t l = scan[p] - scan[sp.first];          // left-goers of this node before p
            const bool left = flag[p] != 0;
            q = left ? sp.first + l : sp.first + sp.left_count + (p - sp.first - l);
            const uint32_t cc = left ? sp.left_count : sp.count - sp.left_count;
            pn = cc > SMALL ? sp.child + (left ? 0u : 1u) : NONE;
        }
    }
    order2[q] = order[p];
    pnode2[q] = pn;
}

// ---------------------------------------------------------------------------------------------- subtrees of <= 64 triangles
// One wave per subtree, a triangle per lane.  Descendant nodes go to the private strip tmp[2 * first ...] with strip-local
// child indices; k_small_emit copies them behind the upper levels.  n_made[k] = nodes made, root_child[k] = strip-local
// index of the root's children (NONE: the root stays a leaf).
#define RT_DSAH_WAVES 4
__global__ void __launch_bounds__(64 * RT_DSAH_WAVES) k_small(const uint32_t* small_list, uint32_t n_small, const rt_bvh_node* out, uint32_t* order,
                                                              const LBox* box, const float* cen, uint32_t max_leaf, float ci, rt_bvh_node* tmp,
                                                              uint32_t* n_made, uint32_t* root_child) {
    __shared__ Bin s_bins[RT_DSAH_WAVES][3 * BINS];
    __shared__ uint32_t s_perm[RT_DSAH_WAVES][64];
    __shared__ uint32_t s_stack[RT_DSAH_WAVES][64][3];            // (strip-local node or NONE for the root, first lane, count)
    __shared__ float s_sbox[RT_DSAH_WAVES][64][6];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * RT_DSAH_WAVES + w;
    if (k >= n_small) return;                                     // (wave-uniform; no workgroup barrier below)
    const uint32_t root = small_list[k];
    const rt_bvh_node rn = out[root];
    const uint32_t first = (uint32_t)rn.left_or_first, count = (uint32_t)rn.prim_count;
    rt_bvh_node* strip = tmp + 2 * (size_t)first;
    Bin* bins = s_bins[w];
    uint32_t id = 0;
    LBox bx; box_reset(bx);
    float c[3] = {0.f, 0.f, 0.f};
    if (lane < count) {
        id = order[first + lane];
        bx = box[id];
        c[0] = cen[(size_t)id * 3]; c[1] = cen[(size_t)id * 3 + 1]; c[2] = cen[(size_t)id * 3 + 2];
    }
    uint32_t made = 0, rchild = NONE, sp = 0;
    if (lane == 0) {
        s_stack[w][0][0] = NONE; s_stack[w][0][1] = 0; s_stack[w][0][2] = count;
        for (int a = 0; a < 3; a++) { s_sbox[w][0][a] = rn.bmin[a]; s_sbox[w][0][3 + a] = rn.bmax[a]; }
    }
    sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp != 0) {
        sp--;
        const uint32_t nloc = s_stack[w][sp][0], nf = s_stack[w][sp][1], nc = s_stack[w][sp][2];
        LBox nb;
        for (int a = 0; a < 3; a++) { nb.mn[a] = s_sbox[w][sp][a]; nb.mx[a] = s_sbox[w][sp][3 + a]; }
        __builtin_amdgcn_wave_barrier();
        const bool in = lane >= nf && lane < nf + nc;
        // centroid bounds and bin scales
        float cmn[3], scale[3];
        for (int a = 0; a < 3; a++) {
            const float lo = wave_min(in ? c[a] : __builtin_inff()), hi = wave_max(in ? c[a] : -__builtin_inff());
            const float ext = hi - lo;
            cmn[a] = lo;
            scale[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
        }
        if (lane < 3 * BINS) {
            Bin e;
            for (int a = 0; a < 3; a++) { e.mn[a] = ORD_PINF; e.mx[a] = ORD_NINF; }
            e.cnt = 0;
            bins[lane] = e;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int mybin[3] = {0, 0, 0};
        if (in) for (int a = 0; a < 3; a++) {
            if (scale[a] == 0.0f) continue;
            mybin[a] = bin_of(c[a], cmn[a], scale[a]);
            bin_add(bins + a * BINS + mybin[a], bx);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // lanes 0..2 scan one axis each; the first axis wins ties (rt_rebuild.h scans the axes in order, strict <)
        float cost = __builtin_inff(); int split = 0;
        const float my_scale = lane == 0 ? scale[0] : (lane == 1 ? scale[1] : scale[2]);
        if (lane < 3 && my_scale != 0.0f) axis_sweep(bins + lane * BINS, cost, split);
        float best_cost = __builtin_inff(); int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float ca = __shfl(cost, a); const int sa = __shfl(split, a);
            if (ca < best_cost) { best_cost = ca; best_axis = a; best_split = sa; }
        }
        uint32_t left_count = 0;
        bool left = false;
        LBox bl, br; box_reset(bl); box_reset(br);
        if (best_axis < 0) {
            if (nc <= max_leaf) continue;                         // leaf (coincident centroids)
            left_count = nc / 2;
            left = in && (lane - nf) < left_count;
        } else {
            if (prefers_leaf(nc, max_leaf, ci, best_cost, nb)) continue;
            left = in && (best_axis == 0 ? mybin[0] : (best_axis == 1 ? mybin[1] : mybin[2])) < best_split;
            left_count = (uint32_t)__popcll(__ballot(left));
        }
        // child boxes: exact unions of the triangles' boxes on either side (what the bins hold, rt_rebuild.h:186)
        for (int a = 0; a < 3; a++) {
            bl.mn[a] = wave_min(left ? bx.mn[a] : __builtin_inff()); bl.mx[a] = wave_max(left ? bx.mx[a] : -__builtin_inff());
            br.mn[a] = wave_min(in && !left ? bx.mn[a] : __builtin_inff()); br.mx[a] = wave_max(in && !left ? bx.mx[a] : -__builtin_inff());
        }
        // stable partition of the lanes [nf, nf + nc): destination lane, inverted through LDS, then the data moves
        const unsigned long long lm = __ballot(left), rm = __ballot(in && !left), below = (1ull << lane) - 1ull;
        const uint32_t dest = !in ? lane : (left ? nf + (uint32_t)__popcll(lm & below) : nf + left_count + (uint32_t)__popcll(rm & below));
        s_perm[w][dest] = lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int src = (int)s_perm[w][lane];
        id = (uint32_t)__shfl((int)id, src);
        for (int a = 0; a < 3; a++) { bx.mn[a] = __shfl(bx.mn[a], src); bx.mx[a] = __shfl(bx.mx[a], src); c[a] = __shfl(c[a], src); }
        // the pair of children, strip-local; the parent becomes interior
        const uint32_t ch = made;
        made += 2;
        if (lane == 0) {
            put_node(strip, ch, bl, (int32_t)(first + nf), (int32_t)left_count);
            put_node(strip, ch + 1, br, (int32_t)(first + nf + left_count), (int32_t)(nc - left_count));
            if (nloc != NONE) { strip[nloc].left_or_first = (int32_t)ch; strip[nloc].prim_count = 0; }
            // right first, then left: the left subtree is numbered first (rt_rebuild.h:190)
            uint32_t s2 = sp;
            if (nc - left_count > 2) {
                s_stack[w][s2][0] = ch + 1; s_stack[w][s2][1] = nf + left_count; s_stack[w][s2][2] = nc - left_count;
                for (int a = 0; a < 3; a++) { s_sbox[w][s2][a] = br.mn[a]; s_sbox[w][s2][3 + a] = br.mx[a]; }
                s2++;
            }
            if (left_count > 2) {
                s_stack[w][s2][0] = ch; s_stack[w][s2][1] = nf; s_stack[w][s2][2] = left_count;
                for (int a = 0; a < 3; a++) { s_sbox[w][s2][a] = bl.mn[a]; s_sbox[w][s2][3 + a] = bl.mx[a]; }
            }
        }
        if (nloc == NONE) rchild = ch;
        sp += (nc - left_count > 2 ? 1u : 0u) + (left_count > 2 ? 1u : 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (lane < count) order[first + lane] = id;
    if (lane == 0) { n_made[k] = made; root_child[k] = rchild; }
}
__global__ void k_small_emit(const uint32_t* small_list, uint32_t n_small, const rt_bvh_node* tmp, const uint32_t* n_made, const uint32_t* base,
                             const uint32_t* root_child, uint32_t nodes_used, rt_bvh_node* out) {
    const uint32_t k = blockIdx.x;
    if (k >= n_small) return;
    const uint32_t root = small_list[k];
    const uint32_t first = (uint32_t)out[root].left_or_first;     // (the root is still a leaf record here: written below by lane 0 only)
    const rt_bvh_node* strip = tmp + 2 * (size_t)first;
    const uint32_t b = nodes_used + base[k], m = n_made[k];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        rt_bvh_node nd = strip[i];
        if (nd.prim_count == 0) nd.left_or_first += (int32_t)b;
        out[b + i] = nd;
    }
    if (threadIdx.x == 0 && root_child[k] != NONE) { out[root].left_or_first = (int32_t)(b + root_child[k]); out[root].prim_count = 0; }
}

#define RT_DSAH_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = hipGetErrorString(e_); return false; } } while (0)

// All mesh BVHs of the scene, on the current device; the result in the same form as rt_rebuild::rebuild.
inline bool rebuild(const rt_scene_desc* s, uint32_t max_leaf, rt_rebuild::MeshLevel& out, const char*& err) {
    out.bvhs.assign(s->mesh_bvhs, s->mesh_bvhs + s->n_mesh_bvhs);
    out.prim_face.assign(s->mesh_prim_face, s->mesh_prim_face + s->n_mesh_prims);
    out.rank.assign(s->n_mesh_prims, 0);
    out.nodes.clear();
    uint32_t n_max = 0;
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) n_max = std::max(n_max, s->mesh_bvhs[b].prim_count);
    if (n_max == 0) { out.nodes.assign(s->mesh_nodes, s->mesh_nodes + s->n_mesh_nodes); return true; }
    float ci = 1.0f;
    if (const char* e = getenv("RT_REBUILD_TRI_COST")) ci = std::max(0.01f, (float)atof(e));

    rt_lbvh::Lap lap;
    rt_lbvh::Scratch d_faces, d_vd, d_work, d_cub;
    RT_DSAH_CHK(d_faces.ensure((size_t)s->n_faces * sizeof(rt_face)));
    RT_DSAH_CHK(d_vd.ensure((size_t)s->n_vertices * sizeof(rt_vertex)));
    RT_DSAH_CHK(hipMemcpy(d_faces.p, s->faces, (size_t)s->n_faces * sizeof(rt_face), hipMemcpyHostToDevice));
    RT_DSAH_CHK(hipMemcpy(d_vd.p, s->vertex_data, (size_t)s->n_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice));
    const size_t N = n_max, F = N / SMALL + 2, S = N / 3 + 2;     // frontier nodes (> 64 triangles each), small subtrees (> 2 each)
    if (2 * F >= (1u << 21)) { err = "device SAH: more than 64 M triangles in one mesh (use the host rebuild)"; return false; }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pf = carve(N * 4), o_box = carve(N * sizeof(LBox)), o_cen = carve(N * 12), o_rb = carve(32), o_ord = carve(N * 4), o_ord2 = carve(N * 4),
                 o_pn = carve(N * 4), o_pn2 = carve(N * 4), o_out = carve(2 * N * sizeof(rt_bvh_node)), o_tmp = carve(2 * N * sizeof(rt_bvh_node)),
                 o_ncb = carve(2 * N * 24), o_nslot = carve(2 * N * 4), o_fa = carve(F * 4), o_fb = carve(F * 4), o_small = carve(S * 4),
                 o_prep = carve(F * sizeof(Prep)), o_bins = carve(F * 3 * BINS * sizeof(Bin)), o_split = carve(F * sizeof(Split)),
                 o_pk = carve(F * 8), o_pks = carve(F * 8), o_flag = carve(N * 4), o_scan = carve(N * 4), o_cnt = carve(32),
                 o_made = carve(S * 4), o_base = carve(S * 4), o_rch = carve(S * 4), o_sf = carve(N * 4);
    RT_DSAH_CHK(d_work.ensure(off));
    char* W = (char*)d_work.p;
    size_t cub_a = 0, cub_b = 0;
    RT_DSAH_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)std::max(N, S)));
    RT_DSAH_CHK(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)F));
    size_t cub_bytes = std::max(cub_a, cub_b);
    RT_DSAH_CHK(d_cub.ensure(cub_bytes));
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
        const int grid_chunk = (int)((n + 256 * CHUNK - 1) / (256 * CHUNK));
        uint32_t* pf = (uint32_t*)(W + o_pf);
        LBox* box = (LBox*)(W + o_box);
        float* cen = (float*)(W + o_cen);
        uint32_t* rb = (uint32_t*)(W + o_rb);
        uint32_t *order = (uint32_t*)(W + o_ord), *order2 = (uint32_t*)(W + o_ord2), *pnode = (uint32_t*)(W + o_pn), *pnode2 = (uint32_t*)(W + o_pn2);
        rt_bvh_node *d_out = (rt_bvh_node*)(W + o_out), *d_tmp = (rt_bvh_node*)(W + o_tmp);
        uint32_t *ncb = (uint32_t*)(W + o_ncb), *nslot = (uint32_t*)(W + o_nslot), *fa = (uint32_t*)(W + o_fa), *fb = (uint32_t*)(W + o_fb);
        uint32_t* small_list = (uint32_t*)(W + o_small);
        Prep* prep = (Prep*)(W + o_prep);
        Bin* bins = (Bin*)(W + o_bins);
        Split* split = (Split*)(W + o_split);
        unsigned long long *pk = (unsigned long long*)(W + o_pk), *pks = (unsigned long long*)(W + o_pks);
        uint32_t *flag = (uint32_t*)(W + o_flag), *scan = (uint32_t*)(W + o_scan), *cnt = (uint32_t*)(W + o_cnt);
        uint32_t *made = (uint32_t*)(W + o_made), *base = (uint32_t*)(W + o_base), *rch = (uint32_t*)(W + o_rch), *sf = (uint32_t*)(W + o_sf);

        RT_DSAH_CHK(hipMemcpy(pf, s->mesh_prim_face + mb.prim_offset, (size_t)n * 4, hipMemcpyHostToDevice));
        const uint32_t rb0[8] = {ORD_PINF, ORD_PINF, ORD_PINF, ORD_NINF, ORD_NINF, ORD_NINF, 0u, 0u};
        RT_DSAH_CHK(hipMemcpy(rb, rb0, 32, hipMemcpyHostToDevice));
        k_prims<<<grid_chunk, 256>>>(pf, (const rt_face*)d_faces.p + mb.face_offset, (const rt_vertex*)d_vd.p, n, box, cen, rb, order, pnode);
        k_root<<<1, 64>>>(rb, n, d_out, ncb, nslot, fa, small_list, cnt);
        uint32_t nodes_used = 1, nf = n > SMALL ? 1u : 0u, n_small = (n > 2 && n <= SMALL) ? 1u : 0u, levels = 0;
        lap("boxes, centroids, root");
        while (nf != 0) {
            levels++;
            const int gf = (int)((nf + 63) / 64);
            k_cbounds<<<grid_chunk, 256>>>(order, pnode, cen, n, ncb);
            k_prep<<<(int)(((size_t)nf * 3 * BINS + 255) / 256), 256>>>(fa, nf, ncb, prep, bins);
            k_bin<<<grid_chunk, 256>>>(order, pnode, nslot, cen, box, prep, n, bins);
            k_split<<<gf, 64>>>(fa, nf, d_out, bins, prep, order, box, max_leaf, ci, split, pk);
            RT_DSAH_CHK(hipcub::DeviceScan::ExclusiveSum(d_cub.p, cub_bytes, pk, pks, (int)nf));
            unsigned long long tot = 0;
            k_totals<<<1, 64>>>(pk, pks, nf, (unsigned long long*)cnt);
            RT_DSAH_CHK(hipMemcpy(&tot, cnt, 8, hipMemcpyDeviceToHost));
            const uint32_t n_split = (uint32_t)(tot & 0x1FFFFFull), n_big = (uint32_t)((tot >> 21) & 0x1FFFFFull), n_sm = (uint32_t)((tot >> 42) & 0x1FFFFFull);
            if (n_big > F || n_small + n_sm > S || nodes_used + 2ull * n_split > 2ull * N) { err = "device SAH: internal bound exceeded"; return false; }
            k_children<<<gf, 64>>>(fa, nf, pks, split, nodes_used, n_small, d_out, ncb, nslot, fb, small_list);
            k_flags<<<grid, 256>>>(order, pnode, nslot, cen, prep, split, n, flag);
            RT_DSAH_CHK(hipcub::DeviceScan::ExclusiveSum(d_cub.p, cub_bytes, flag, scan, (int)n));
            k_scatter<<<grid, 256>>>(order, pnode, nslot, split, flag, scan, n, order2, pnode2);
            std::swap(order, order2);
            std::swap(pnode, pnode2);
            std::swap(fa, fb);
            nodes_used += 2u * n_split;
            n_small += n_sm;
            nf = n_big;
            if (levels > 4096) { err = "device SAH: too many levels"; return false; }
        }
        RT_DSAH_CHK(hipGetLastError());
        lap("upper levels");
        uint32_t n_nodes = nodes_used;
        if (n_small != 0) {
            k_small<<<(int)((n_small + RT_DSAH_WAVES - 1) / RT_DSAH_WAVES), 64 * RT_DSAH_WAVES>>>(small_list, n_small, d_out, order, box, cen, max_leaf, ci, d_tmp, made, rch);
            RT_DSAH_CHK(hipcub::DeviceScan::ExclusiveSum(d_cub.p, cub_bytes, made, base, (int)n_small));
            uint32_t lastv[2];
            RT_DSAH_CHK(hipMemcpy(&lastv[0], made + (n_small - 1), 4, hipMemcpyDeviceToHost));
            RT_DSAH_CHK(hipMemcpy(&lastv[1], base + (n_small - 1), 4, hipMemcpyDeviceToHost));
            if ((uint64_t)nodes_used + lastv[0] + lastv[1] > 2ull * N) { err = "device SAH: internal bound exceeded"; return false; }
            k_small_emit<<<(int)n_small, 64>>>(small_list, n_small, d_tmp, made, base, rch, nodes_used, d_out);
            n_nodes = nodes_used + lastv[0] + lastv[1];
        }
        rt_lbvh::k_sorted_faces<<<grid, 256>>>(order, pf, n, sf);
        RT_DSAH_CHK(hipGetLastError());
        lap("subtrees of <= 64");
        out.nodes.resize(node_off + n_nodes);
        RT_DSAH_CHK(hipMemcpy(out.nodes.data() + node_off, d_out, (size_t)n_nodes * sizeof(rt_bvh_node), hipMemcpyDeviceToHost));
        // leaf order of the new tree, and each slot's position in the caller's leaf order — which is what `order` holds
        RT_DSAH_CHK(hipMemcpy(out.prim_face.data() + mb.prim_offset, sf, (size_t)n * 4, hipMemcpyDeviceToHost));
        uint32_t* rk = out.rank.data() + mb.prim_offset;
        RT_DSAH_CHK(hipMemcpy(rk, order, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (mb.prim_offset != 0) for (uint32_t p = 0; p < n; p++) rk[p] += mb.prim_offset;
        mb.node_offset = (uint32_t)node_off;
        mb.node_count = n_nodes;
        lap("download + rank");
    }
    return true;
}
#undef RT_DSAH_CHK

}  // namespace rt_dsah
