// Host test of raytracer_amd/csrc/rt_rebuild.h (the SAH rebuild behind RT_UPLOAD_REBUILD_MESH_BVH): the trees it
// builds must be valid input for the upload path — reference node format, every face in exactly one leaf, boxes
// that hold their triangles and their children, children stored after their parent, left range before right range
// — and the rank array must be the caller's leaf positions.  Soups, grids, coincident centroids, 0..3 triangles.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../raytracer_amd/csrc/rt_rebuild.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

struct Mesh { std::vector<rt_vertex> vd; std::vector<rt_face> faces; };

static void add_tri(Mesh& m, const float* a, const float* b, const float* c) {
    const float* p[3] = {a, b, c};
    rt_face f{};
    int32_t* idx[3] = {&f.i0, &f.i1, &f.i2};
    for (int k = 0; k < 3; k++) {
        rt_vertex v{};
        for (int j = 0; j < 3; j++) v.pos[j] = p[k][j];
        m.vd.push_back(v);
        *idx[k] = (int32_t)m.vd.size();
    }
    m.faces.push_back(f);
}

static void check_scene(const char* name, const std::vector<Mesh>& meshes, uint32_t max_leaf, std::mt19937& rng) {
    // concatenate like the flattener: one BVH per mesh, caller's leaf order = a random permutation
    std::vector<rt_vertex> vd; std::vector<rt_face> faces; std::vector<uint32_t> prim_face; std::vector<rt_mesh_bvh> bvhs;
    std::vector<rt_bvh_node> dummy_nodes;
    for (const Mesh& m : meshes) {
        rt_mesh_bvh mb{};
        mb.node_offset = (uint32_t)dummy_nodes.size(); mb.node_count = m.faces.empty() ? 0 : 1;
        if (!m.faces.empty()) dummy_nodes.push_back(rt_bvh_node{});
        mb.prim_offset = (uint32_t)prim_face.size(); mb.prim_count = (uint32_t)m.faces.size();
        mb.face_offset = (uint32_t)faces.size(); mb.face_count = (uint32_t)m.faces.size();
        const int32_t vbase = (int32_t)vd.size();
        vd.insert(vd.end(), m.vd.begin(), m.vd.end());
        for (rt_face f : m.faces) { f.i0 += vbase; f.i1 += vbase; f.i2 += vbase; faces.push_back(f); }
        std::vector<uint32_t> perm(m.faces.size());
        for (uint32_t i = 0; i < perm.size(); i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        prim_face.insert(prim_face.end(), perm.begin(), perm.end());
        bvhs.push_back(mb);
    }
    rt_scene_desc s{};
    s.mesh_bvhs = bvhs.data(); s.n_mesh_bvhs = (uint32_t)bvhs.size();
    s.mesh_nodes = dummy_nodes.data(); s.n_mesh_nodes = (uint32_t)dummy_nodes.size();
    s.mesh_prim_face = prim_face.data(); s.n_mesh_prims = (uint32_t)prim_face.size();
    s.faces = faces.data(); s.n_faces = (uint32_t)faces.size();
    s.vertex_data = vd.data(); s.n_vertices = (uint32_t)vd.size();
    rt_rebuild::MeshLevel R;
    rt_rebuild::rebuild(&s, max_leaf, R);
    CHECK(R.prim_face.size() == prim_face.size() && R.rank.size() == prim_face.size(), "%s: sizes", name);
    size_t total_nodes = 0, max_leaf_seen = 0, max_depth = 0;
    for (size_t b = 0; b < bvhs.size(); b++) {
        const rt_mesh_bvh& mb = R.bvhs[b];
        CHECK(mb.prim_offset == bvhs[b].prim_offset && mb.prim_count == bvhs[b].prim_count && mb.face_offset == bvhs[b].face_offset, "%s: bvh %zu header", name, b);
        CHECK(mb.node_offset == total_nodes, "%s: bvh %zu node offset", name, b);
        total_nodes += mb.node_count;
        if (mb.prim_count == 0) continue;
        const rt_bvh_node* nd = R.nodes.data() + mb.node_offset;
        const uint32_t* pf = R.prim_face.data() + mb.prim_offset;
        // permutation of the faces, ranks = caller's positions
        std::vector<int> seen(mb.face_count, 0);
        for (uint32_t p = 0; p < mb.prim_count; p++) {
            CHECK(pf[p] < mb.face_count, "%s: face index", name);
            seen[pf[p]]++;
            const uint32_t r = R.rank[mb.prim_offset + p];
            CHECK(r >= mb.prim_offset && r < mb.prim_offset + mb.prim_count && prim_face[r] == pf[p], "%s: rank of slot %u", name, p);
        }
        for (int c : seen) CHECK(c == 1, "%s: a face is in %d leaves", name, c);
        // walk: DFS left first; leaves must tile [0, prim_count) in that order
        struct It { uint32_t n; size_t depth; };
        std::vector<It> st{{0, 1}};
        uint32_t next = 0, visited = 0;
        while (!st.empty()) {
            It it = st.back(); st.pop_back();
            const rt_bvh_node& n = nd[it.n];
            visited++;
            max_depth = std::max(max_depth, it.depth);
            if (n.prim_count > 0) {
                CHECK((uint32_t)n.left_or_first == next, "%s: leaf ranges out of order", name);
                next += (uint32_t)n.prim_count;
                max_leaf_seen = std::max(max_leaf_seen, (size_t)n.prim_count);
                for (int32_t k = 0; k < n.prim_count; k++) {
                    const rt_face& f = faces[mb.face_offset + pf[n.left_or_first + k]];
                    for (int32_t vi : {f.i0, f.i1, f.i2})
                        for (int a = 0; a < 3; a++)
                            CHECK(vd[vi - 1].pos[a] >= n.bmin[a] && vd[vi - 1].pos[a] <= n.bmax[a], "%s: vertex outside its leaf box", name);
                }
            } else {
                CHECK(n.prim_count == 0 && n.left_or_first > (int32_t)it.n && (uint32_t)n.left_or_first + 1 < mb.node_count, "%s: children of node %u", name, it.n);
                for (int c = 0; c < 2; c++) {
                    const rt_bvh_node& ch = nd[n.left_or_first + c];
                    for (int a = 0; a < 3; a++) CHECK(ch.bmin[a] >= n.bmin[a] && ch.bmax[a] <= n.bmax[a], "%s: child box outside parent", name);
                }
                st.push_back({(uint32_t)n.left_or_first + 1, it.depth + 1});
                st.push_back({(uint32_t)n.left_or_first, it.depth + 1});
            }
        }
        CHECK(next == mb.prim_count, "%s: leaves cover %u of %u prims", name, next, mb.prim_count);
        CHECK(visited == mb.node_count, "%s: %u of %u nodes reachable", name, visited, mb.node_count);
    }
    CHECK(total_nodes == R.nodes.size(), "%s: node total", name);
    printf("%-28s max_leaf %u: %zu nodes, largest leaf %zu, depth %zu\n", name, max_leaf, total_nodes, max_leaf_seen, max_depth);
    CHECK(max_depth <= 64, "%s: deeper than the reference's stack", name);
}

int main() {
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (uint32_t max_leaf : {2u, 4u, 8u}) {
        {   // random soup of small triangles + two tiny meshes + an empty one
            Mesh soup;
            for (int i = 0; i < 20000; i++) {
                float c[3] = {U(rng) * 10, U(rng) * 10, U(rng) * 10}, a[3], b[3], d[3];
                for (int k = 0; k < 3; k++) { a[k] = c[k] + U(rng) * .2f; b[k] = c[k] + U(rng) * .2f; d[k] = c[k] + U(rng) * .2f; }
                add_tri(soup, a, b, d);
            }
            Mesh one, three, none;
            float p0[3] = {0, 0, 0}, p1[3] = {1, 0, 0}, p2[3] = {0, 1, 0}, p3[3] = {0, 0, 1};
            add_tri(one, p0, p1, p2);
            add_tri(three, p0, p1, p2); add_tri(three, p0, p1, p3); add_tri(three, p0, p2, p3);
            check_scene("soup + tiny + empty", {soup, one, none, three}, max_leaf, rng);
        }
        if (max_leaf == 4) {   // large enough for the parallel binning of the upper levels (>= 65536 triangles)
            Mesh big;
            for (int i = 0; i < 150000; i++) {
                float c[3] = {U(rng) * 30, U(rng) * 3, U(rng) * 10}, a[3], b[3], d[3];
                for (int k = 0; k < 3; k++) { a[k] = c[k] + U(rng) * .1f; b[k] = c[k] + U(rng) * .1f; d[k] = c[k] + U(rng) * .1f; }
                add_tri(big, a, b, d);
            }
            check_scene("large soup", {big}, max_leaf, rng);
        }
        {   // regular grid (the bench mesh's kind), long thin extent
            Mesh grid;
            const int W = 300, H = 40;
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
                float a[3] = {(float)x, (float)y, 0}, b[3] = {(float)x + 1, (float)y, 0}, c[3] = {(float)x, (float)y + 1, 0}, d[3] = {(float)x + 1, (float)y + 1, 0.1f};
                add_tri(grid, a, b, c); add_tri(grid, b, d, c);
            }
            check_scene("grid", {grid}, max_leaf, rng);
        }
        {   // 5000 triangles with one and the same centroid (the reference leaves them in one leaf), plus duplicates
            Mesh same;
            for (int i = 0; i < 5000; i++) {
                float r = 1.0f + (float)(i % 50), a[3] = {r, 0, 0}, b[3] = {-r * .5f, r, 0}, c[3] = {-r * .5f, -r, 0};
                add_tri(same, a, b, c);
            }
            check_scene("coincident centroids", {same}, max_leaf, rng);
        }
    }
    printf("total: %d failures\n", fails);
    return fails ? 1 : 0;
}
