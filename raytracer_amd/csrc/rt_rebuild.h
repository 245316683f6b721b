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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "rt.h"

namespace rt_rebuild {

struct MeshLevel {
    std::vector<rt_bvh_node> nodes;        // all BVHs, concatenated
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

// One BVH over the faces listed in prim_face[0..n) (indices relative to `faces`).  Appends its nodes to `nodes`
// (node 0 of the BVH first) and reorders prim_face in place.
inline void build_one(const rt_face* faces, const rt_vertex* vd, uint32_t* prim_face, uint32_t n, uint32_t max_leaf,
                      std::vector<rt_bvh_node>& nodes) {
    const size_t base = nodes.size();
    if (n == 0) return;
    // per face of this mesh (indexed like prim_face's values): bounds and centroid
    uint32_t max_face = 0;
    for (uint32_t i = 0; i < n; i++) max_face = std::max(max_face, prim_face[i]);
    std::vector<Box> fb(max_face + 1);
    std::vector<float> fc((size_t)(max_face + 1) * 3);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t f = prim_face[i];
        const float* a = vd[faces[f].i0 - 1].pos;
        const float* b = vd[faces[f].i1 - 1].pos;
        const float* c = vd[faces[f].i2 - 1].pos;
        fb[f].reset(); fb[f].add(a); fb[f].add(b); fb[f].add(c);
        for (int k = 0; k < 3; k++) fc[(size_t)f * 3 + k] = (a[k] + b[k] + c[k]) * (1.0f / 3.0f);
    }
    auto set_bounds = [&](rt_bvh_node& nd, uint32_t first, uint32_t count) {
        Box b; b.reset();
        for (uint32_t i = 0; i < count; i++) b.add(fb[prim_face[first + i]]);
        std::memcpy(nd.bmin, b.mn, 12); std::memcpy(nd.bmax, b.mx, 12);
    };
    nodes.reserve(base + 2ull * n);
    nodes.push_back(rt_bvh_node{});
    nodes[base].left_or_first = 0; nodes[base].prim_count = (int32_t)n;
    set_bounds(nodes[base], 0, n);
    std::vector<uint32_t> todo{0};
    while (!todo.empty()) {
        const uint32_t ni = todo.back(); todo.pop_back();
        const uint32_t first = (uint32_t)nodes[base + ni].left_or_first, count = (uint32_t)nodes[base + ni].prim_count;
        if (count <= 2) continue;
        // centroid bounds
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < count; i++) {
            const float* c = &fc[(size_t)prim_face[first + i] * 3];
            for (int a = 0; a < 3; a++) { cmn[a] = std::min(cmn[a], c[a]); cmx[a] = std::max(cmx[a], c[a]); }
        }
        float best_cost = INFINITY; int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cmx[a] - cmn[a];
            if (!(ext > 0.0f)) continue;
            const float scale = (float)BINS / ext;
            Box bb[BINS]; uint32_t bc[BINS];
            for (int k = 0; k < BINS; k++) { bb[k].reset(); bc[k] = 0; }
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t f = prim_face[first + i];
                int k = (int)((fc[(size_t)f * 3 + a] - cmn[a]) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                bb[k].add(fb[f]); bc[k]++;
            }
            float la[BINS - 1]; uint32_t lc[BINS - 1];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int k = 0; k < BINS - 1; k++) { if (bc[k]) acc.add(bb[k]); cnt += bc[k]; la[k] = cnt ? acc.half_area() : 0.0f; lc[k] = cnt; }
            acc.reset(); cnt = 0;
            for (int k = BINS - 1; k > 0; k--) {
                if (bc[k]) acc.add(bb[k]); cnt += bc[k];
                if (cnt == 0 || lc[k - 1] == 0) continue;
                const float cost = la[k - 1] * (float)lc[k - 1] + acc.half_area() * (float)cnt;
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
        }
        uint32_t left_count;
        if (best_axis < 0) {
            // all centroids coincide: the reference stops here (raytracer.cpp:317) and leaves one big leaf; halve it
            if (count <= max_leaf) continue;
            left_count = count / 2;
        } else {
            // leaf instead of a split when no partition is cheaper than testing the triangles (leaf cost 1 per
            // triangle, node cost 1)
            if (count <= max_leaf) {
                Box nb; std::memcpy(nb.mn, nodes[base + ni].bmin, 12); std::memcpy(nb.mx, nodes[base + ni].bmax, 12);
                const float pa = nb.half_area();
                if (!(pa > 0.0f) || best_cost / pa + 1.0f >= (float)count) continue;
            }
            const float scale = (float)BINS / (cmx[best_axis] - cmn[best_axis]);
            uint32_t i = first, j = first + count;
            while (i < j) {
                int k = (int)((fc[(size_t)prim_face[i] * 3 + best_axis] - cmn[best_axis]) * scale);
                k = k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
                if (k < best_split) i++; else std::swap(prim_face[i], prim_face[--j]);
            }
            left_count = i - first;
            if (left_count == 0 || left_count == count) left_count = count / 2;   // (cannot happen: both sides were non-empty in the bins)
        }
        const uint32_t l = (uint32_t)(nodes.size() - base);
        nodes.push_back(rt_bvh_node{}); nodes.push_back(rt_bvh_node{});
        rt_bvh_node& L = nodes[base + l]; rt_bvh_node& R = nodes[base + l + 1];
        L.left_or_first = (int32_t)first; L.prim_count = (int32_t)left_count;
        R.left_or_first = (int32_t)(first + left_count); R.prim_count = (int32_t)(count - left_count);
        set_bounds(L, first, left_count); set_bounds(R, first + left_count, count - left_count);
        nodes[base + ni].left_or_first = (int32_t)l; nodes[base + ni].prim_count = 0;
        todo.push_back(l + 1); todo.push_back(l);      // left subtree first: node order close to the reference's
    }
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
