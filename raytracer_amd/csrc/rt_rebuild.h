// rt_rebuild.h — host side of RT_UPLOAD_REBUILD_MESH_BVH (include/rt.h): a binned-SAH rebuild of the mesh BVHs.
//
// NON-PARITY mode, SURVEY §8f-3.  The reference splits every node at the midpoint of its longest axis
// (SubdivideMesh, raytracer.cpp:296-339) until two triangles are left; on scanned meshes that gives deep,
// unbalanced trees and, where centroids coincide, leaves of thousands of triangles (blog/hw2/episode2.md:240-248).
// This builder keeps the reference's node format (rt_bvh_node: children adjacent, leaves as ranges of a prim
// array, left range before right range) so that everything downstream — validation, the device-side collapse
// into 4-wide records, the traversal kernels — is the code of the parity path; only the tree differs.  A
// different tree visits triangles in a different order, so scenes uploaded this way are always traced with the
// near-first kernels (rt_traverse.h, NEAR), and ties between equal t are settled by each triangle's position in
// the REFERENCE's leaf order, kept as a side array (`rank`): the closest hit is then the one the reference
// returns whenever its box culls agree with its triangle tests.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <vector>
#include "rt.h"

namespace rt_rebuild {

// std::allocator whose value-less construct() default-initialises: resize() of a vector of PODs then does not write the
// memory first (the device builders resize to a tree's node count and copy 35 MB per million triangles over it)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
using NodeVec = std::vector<rt_bvh_node, NoInitAlloc<rt_bvh_node>>;

struct MeshLevel {
    NodeVec nodes;                         // all BVHs, concatenated
    std::vector<rt_mesh_bvh> bvhs;         // node_offset / node_count rewritten, the rest as given
    std::vector<uint32_t> prim_face;       // leaf order of the rebuilt trees (same slots per BVH as the caller's array)
    std::vector<uint32_t> rank;            // per slot: position of that face in the caller's (the reference's) leaf order
};

struct Box {
    float mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
    void add(const Box& b) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    void add(const float* p) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    float half_area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int BINS = 16;

inline unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("RT_REBUILD_THREADS")) n = (unsigned)std::max(1, atoi(e));
    return std::max(1u, std::min(n, 16u));
}

// fn(t, T) for every share t of T, on T threads (T == 1: inline).  A share whose thread cannot be started runs on
// the calling thread; an exception in any share is rethrown here as std::bad_alloc after all threads have joined
// (nothing may unwind out of a std::thread).
template <class F>
inline void on_threads(unsigned T, F fn) {
    if (T <= 1) { fn(0u, 1u); return; }
    std::atomic<bool> failed{false};
    auto run = [&](unsigned t) { try { fn(t, T); } catch (...) { failed = true; } };
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        th.reserve(T - 1);
        for (; started < T; started++) th.emplace_back(run, started);
    } catch (...) {}
    run(0u);
    for (unsigned t = started; t < T; t++) run(t);
    for (auto& x : th) x.join();
    if (failed) throw std::bad_alloc();
}

struct Bins {
    Box bb[3][BINS]; uint32_t bc[3][BINS];
    float cmn[3], cmx[3];
    void reset() { for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++) { bb[a][k].reset(); bc[a][k] = 0; } }
};

struct Builder {
    const Box* fb;              // per face: bounds
    const float* fc;            // per face: centroid
    uint32_t* prim_face;        // this BVH's leaf order (reordered in place)
    uint32_t max_leaf;
    float ci;                   // cost of one triangle test in units of one node step (the leaf-or-split decision)

    static int bin_of(float c, float lo, float scale) {
        int k = (int)((c - lo) * scale);
        return k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
    }

    // Splits nodes[root] (a leaf on entry) and its descendants; new nodes are appended to `nodes`, children
    // adjacent and after their parent, left subtree first.  Leaves of at most `defer_below` triangles that would
    // still be split are listed in `deferred` instead (their subtrees are grown by other threads).  Nodes of
    // `par_above` triangles or more are binned by `T` threads.
    void grow(std::vector<rt_bvh_node>& nodes, uint32_t root, uint32_t defer_below, std::vector<uint32_t>* deferred,
              uint32_t par_above, unsigned T) const {
        std::vector<uint32_t> todo{root};
        std::vector<Bins> part(1);
        while (!todo.empty()) {
            const uint32_t ni = todo.back(); todo.pop_back();
            const uint32_t first = (uint32_t)nodes[ni].left_or_first, count = (uint32_t)nodes[ni].prim_count;
            if (count <= 2) continue;
            if (deferred && count <= defer_below) { deferred->push_back(ni); continue; }
            const unsigned nt = (count >= par_above && T > 1) ? T : 1;
            if (part.size() < nt) part.resize(nt);
            // centroid bounds, then the bins of all three axes in one pass
            on_threads(nt, [&](unsigned t, unsigned TT) {
                Bins& B = part[t];
                for (int a = 0; a < 3; a++) { B.cmn[a] = INFINITY; B.cmx[a] = -INFINITY; }
                const uint32_t lo = first + (uint32_t)((uint64_t)count * t / TT), hi = first + (uint32_t)((uint64_t)count * (t + 1) / TT);
                for (uint32_t i = lo; i < hi; i++) {
                    const float* c = &fc[(size_t)prim_face[i] * 3];
                    for (int a = 0; a < 3; a++) { B.cmn[a] = std::min(B.cmn[a], c[a]); B.cmx[a] = std::max(B.cmx[a], c[a]); }
                }
            });
            float cmn[3], cmx[3], scale[3];
            for (int a = 0; a < 3; a++) {
                cmn[a] = part[0].cmn[a]; cmx[a] = part[0].cmx[a];
                for (unsigned t = 1; t < nt; t++) { cmn[a] = std::min(cmn[a], part[t].cmn[a]); cmx[a] = std::max(cmx[a], part[t].cmx[a]); }
                const float ext = cmx[a] - cmn[a];
                scale[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
            }
            on_threads(nt, [&](unsigned t, unsigned TT) {
                Bins& B = part[t];
                B.reset();
                const uint32_t lo = first + (uint32_t)((uint64_t)count * t / TT), hi = first + (uint32_t)((uint64_t)count * (t + 1) / TT);
                for (uint32_t i = lo; i < hi; i++) {
                    const uint32_t f = prim_face[i];
                    for (int a = 0; a < 3; a++) {
                        if (scale[a] == 0.0f) continue;
                        const int k = bin_of(fc[(size_t)f * 3 + a], cmn[a], scale[a]);
                        B.bb[a][k].add(fb[f]); B.bc[a][k]++;
                    }
                }
            });
            Bins& B = part[0];
            for (unsigned t = 1; t < nt; t++)
                for (int a = 0; a < 3; a++) for (int k = 0; k < BINS; k++) if (part[t].bc[a][k]) { B.bb[a][k].add(part[t].bb[a][k]); B.bc[a][k] += part[t].bc[a][k]; }
            float best_cost = INFINITY; int best_axis = -1, best_split = 0;
            Box best_l, best_r; best_l.reset(); best_r.reset();
            for (int a = 0; a < 3; a++) {
                if (scale[a] == 0.0f) continue;
                float la[BINS - 1]; uint32_t lc[BINS - 1]; Box lb[BINS - 1];
                Box acc; acc.reset(); uint32_t cnt = 0;
                for (int k = 0; k < BINS - 1; k++) {
                    if (B.bc[a][k]) acc.add(B.bb[a][k]);
                    cnt += B.bc[a][k];
                    la[k] = cnt ? acc.half_area() : 0.0f; lc[k] = cnt; lb[k] = acc;
                }
                acc.reset(); cnt = 0;
                for (int k = BINS - 1; k > 0; k--) {
                    if (B.bc[a][k]) acc.add(B.bb[a][k]);
                    cnt += B.bc[a][k];
                    if (cnt == 0 || lc[k - 1] == 0) continue;
                    const float cost = la[k - 1] * (float)lc[k - 1] + acc.half_area() * (float)cnt;
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; best_l = lb[k - 1]; best_r = acc; }
                }
            }
            uint32_t left_count;
            if (best_axis < 0) {
                // all centroids coincide: the reference stops here (raytracer.cpp:317) and leaves one big leaf; halve it
                if (count <= max_leaf) continue;
                left_count = count / 2;
                best_l.reset(); best_r.reset();
                for (uint32_t i = 0; i < count; i++) (i < left_count ? best_l : best_r).add(fb[prim_face[first + i]]);
            } else {
                // leaf instead of a split when no partition is cheaper than testing the triangles (leaf cost 1 per
                // triangle, node cost 1)
                if (count <= max_leaf) {
                    Box nb; std::memcpy(nb.mn, nodes[ni].bmin, 12); std::memcpy(nb.mx, nodes[ni].bmax, 12);
                    const float pa = nb.half_area();
                    if (!(pa > 0.0f) || ci * best_cost / pa + 1.0f >= ci * (float)count) continue;
                }
                uint32_t i = first, j = first + count;
                while (i < j) {
                    if (bin_of(fc[(size_t)prim_face[i] * 3 + best_axis], cmn[best_axis], scale[best_axis]) < best_split) i++;
                    else std::swap(prim_face[i], prim_face[--j]);
                }
                left_count = i - first;       // (> 0 and < count: both sides of the split held triangles in the bins)
            }
            const uint32_t l = (uint32_t)nodes.size();
            nodes.push_back(rt_bvh_node{}); nodes.push_back(rt_bvh_node{});
            rt_bvh_node& L = nodes[l]; rt_bvh_node& R = nodes[l + 1];
            L.left_or_first = (int32_t)first; L.prim_count = (int32_t)left_count;
            R.left_or_first = (int32_t)(first + left_count); R.prim_count = (int32_t)(count - left_count);
            // the bins hold exact unions of triangle boxes, so these are the boxes UpdateMeshNodeBounds would give
            std::memcpy(L.bmin, best_l.mn, 12); std::memcpy(L.bmax, best_l.mx, 12);
            std::memcpy(R.bmin, best_r.mn, 12); std::memcpy(R.bmax, best_r.mx, 12);
            nodes[ni].left_or_first = (int32_t)l; nodes[ni].prim_count = 0;
            todo.push_back(l + 1); todo.push_back(l);      // left subtree first
        }
    }
};

// One BVH over the faces listed in prim_face[0..n) (indices relative to `faces`).  Appends its nodes to `out`
// (node 0 of the BVH first) and reorders prim_face in place.  The upper levels are grown by this thread with
// parallel binning; subtrees of a few thousand triangles are grown by all threads side by side and spliced in.
inline void build_one(const rt_face* faces, const rt_vertex* vd, uint32_t* prim_face, uint32_t n, uint32_t max_leaf,
                      NodeVec& out) {
    if (n == 0) return;
    const unsigned T = n >= 16384 ? host_threads() : 1;       // (small meshes: starting threads costs more than the build)
    uint32_t max_face = 0;
    for (uint32_t i = 0; i < n; i++) max_face = std::max(max_face, prim_face[i]);
    std::vector<Box> fb(max_face + 1);
    std::vector<float> fc((size_t)(max_face + 1) * 3);
    on_threads(n >= 65536 ? T : 1, [&](unsigned t, unsigned TT) {
        const uint32_t lo = (uint32_t)((uint64_t)n * t / TT), hi = (uint32_t)((uint64_t)n * (t + 1) / TT);
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t f = prim_face[i];
            const float* a = vd[faces[f].i0 - 1].pos;
            const float* b = vd[faces[f].i1 - 1].pos;
            const float* c = vd[faces[f].i2 - 1].pos;
            fb[f].reset(); fb[f].add(a); fb[f].add(b); fb[f].add(c);
            for (int k = 0; k < 3; k++) fc[(size_t)f * 3 + k] = (a[k] + b[k] + c[k]) * (1.0f / 3.0f);
        }
    });
    float ci = 1.0f;
    if (const char* e = getenv("RT_REBUILD_TRI_COST")) ci = std::max(0.01f, (float)atof(e));
    Builder bld{fb.data(), fc.data(), prim_face, max_leaf, ci};
    std::vector<rt_bvh_node> nodes;
    nodes.reserve(2ull * n);
    nodes.push_back(rt_bvh_node{});
    nodes[0].left_or_first = 0; nodes[0].prim_count = (int32_t)n;
    {
        Box b; b.reset();
        for (uint32_t i = 0; i < n; i++) b.add(fb[prim_face[i]]);
        std::memcpy(nodes[0].bmin, b.mn, 12); std::memcpy(nodes[0].bmax, b.mx, 12);
    }
    std::vector<uint32_t> deferred;
    const uint32_t defer_below = T > 1 ? std::max(2048u, n / (16u * T)) : 0u;
    bld.grow(nodes, 0, defer_below, T > 1 ? &deferred : nullptr, 65536u, T);
    if (!deferred.empty()) {
        std::vector<std::vector<rt_bvh_node>> sub(deferred.size());
        std::atomic<uint32_t> next{0};
        on_threads(T, [&](unsigned, unsigned) {
            for (;;) {
                const uint32_t k = next.fetch_add(1);
                if (k >= deferred.size()) break;
                sub[k].push_back(nodes[deferred[k]]);
                bld.grow(sub[k], 0, 0, nullptr, 0xFFFFFFFFu, 1);
            }
        });
        for (size_t k = 0; k < deferred.size(); k++) {
            const std::vector<rt_bvh_node>& loc = sub[k];
            const int32_t off = (int32_t)nodes.size() - 1;           // local node i >= 1 becomes node off + i
            rt_bvh_node root = loc[0];
            if (root.prim_count == 0) root.left_or_first += off;
            nodes[deferred[k]] = root;
            for (size_t i = 1; i < loc.size(); i++) {
                rt_bvh_node nd = loc[i];
                if (nd.prim_count == 0) nd.left_or_first += off;
                nodes.push_back(nd);
            }
        }
    }
    out.insert(out.end(), nodes.begin(), nodes.end());
}

inline void rebuild(const rt_scene_desc* s, uint32_t max_leaf, MeshLevel& out) {
    out.bvhs.assign(s->mesh_bvhs, s->mesh_bvhs + s->n_mesh_bvhs);
    out.prim_face.assign(s->mesh_prim_face, s->mesh_prim_face + s->n_mesh_prims);
    out.rank.assign(s->n_mesh_prims, 0);
    out.nodes.clear();
    std::vector<uint32_t> ref_pos;
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
        rt_mesh_bvh& mb = out.bvhs[b];
        const size_t off = out.nodes.size();
        if (mb.prim_count == 0) {                  // an empty mesh keeps whatever nodes the caller gave it
            out.nodes.insert(out.nodes.end(), s->mesh_nodes + mb.node_offset, s->mesh_nodes + mb.node_offset + mb.node_count);
            mb.node_offset = (uint32_t)off;
            continue;
        }
        uint32_t* pf = out.prim_face.data() + mb.prim_offset;
        ref_pos.assign(mb.face_count, 0);
        for (uint32_t p = 0; p < mb.prim_count; p++) ref_pos[pf[p]] = p;
        build_one(s->faces + mb.face_offset, s->vertex_data, pf, mb.prim_count, max_leaf, out.nodes);
        for (uint32_t p = 0; p < mb.prim_count; p++) out.rank[mb.prim_offset + p] = mb.prim_offset + ref_pos[pf[p]];
        mb.node_offset = (uint32_t)off;
        mb.node_count = (uint32_t)(out.nodes.size() - off);
    }
}

}  // namespace rt_rebuild
