// rt_hip.hip — C ABI (include/rt.h) of the gfx950 ray path: scene upload
// (re-layout of the flattened reference scene into the device layout of
// rt_device.h) and the render driver that enqueues the wavefront pipeline of
// rt_kernels.h.  Host code here is set-up and launch logic only; there is no
// CPU rendering path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

#include "rt.h"
#include "rt_kernels.h"
#include "rt_ordered.h"
#include "rt_mega.h"
#include "rt_upload.h"
#include "rt_rebuild.h"
#include "rt_lbvh.h"
#include "rt_dsah.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                               \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return RT_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (need == 0) return RT_OK;
        HIPCHK(hipMalloc(&p, need));
        bytes = need;
        return RT_OK;
    }
    // grow to `need` bytes, keeping the first `keep` bytes (device-to-device copy)
    int grow_keep(size_t need, size_t keep) {
        if (need <= bytes) return RT_OK;
        void* q = nullptr;
        HIPCHK(hipMalloc(&q, need));
        if (p && keep) {
            hipError_t e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { (void)hipFree(q); return fail(RT_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e)); }
        }
        if (p) (void)hipFree(p);
        p = q;
        bytes = need;
        return RT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

template <class T>
int upload_vec(DevBuf& b, const std::vector<T>& v) {
    int rc = b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

enum TimeCat { TC_RAYGEN, TC_CLOSEST, TC_SHADE, TC_SHADOW, TC_COMBINE, TC_RESOLVE, TC_COUNT };

}  // namespace

// ---- work memory of rt_render_tile_device_noalloc (include/rt.h): one array per GPU, inside the code object, so
// that the call needs no hipMalloc.  It holds everything a render otherwise keeps in buffers of the scene: the
// Control block, the traversal spill stacks and donation candidates of a reduced trace grid, the ray-record pool
// and shadow slots of a chunk, and the MT19937 scratch of multi-sample chunks.  `scratch` itself is only a
// host-side handle of the symbol: the device address of each GPU's copy is asked for once per device, with that
// device current (rt_upload_scene_ex), and only that pointer is ever handed to kernels, memsets or copies.
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU race (rt.h says so).
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
#define RT_NOALLOC_MAX_SAMPLES (1u << 22)   // tile_w * tile_h * num_samples accepted by the entry point
#define RT_NOALLOC_BLOCKS 16                // trace workgroups it launches (their spill / candidate areas must fit)
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES];   // device address of `scratch` on each GPU (0 = not asked yet)
static std::mutex g_scratch_mutex;

struct MeshRoot { uint32_t root_link; float rmin[3], rmax[3]; };
struct rt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    int blocks_closest = 0, blocks_shadow = 0, blocks_ordered = 0;   // persistent grid sizes

    // device scene
    DevBuf b_big, b_nodes, b_tris, b_tri_shade, b_vnormals, b_objects, b_top_prims, b_top_items, b_planes, b_materials, b_lights, b_area_lights;
    DScene ds{};
    DevBuf b_tri_rank;                // rebuilt mesh BVHs only (rt_rebuild.h)
    bool rebuilt = false;             // uploaded with RT_UPLOAD_REBUILD_MESH_BVH: always traced nearest-first
    bool has_reflective = false;      // any material with type != None
    bool needs_sequential_rng = false;   // area lights or rough reflective materials (RNG drawn while shading)
    uint32_t n_point_lights = 0, n_area_lights = 0;
    // what rt_update_scene keeps: the mesh level as laid out by rt_upload_scene
    struct MeshLevel {
        std::vector<struct MeshRoot> roots;        // per mesh BVH: root link + root box
        std::vector<DBigLeaf> big;                  // big leaves of the mesh BVHs
        uint32_t n_node_records = 0;                // DNode4 records of the mesh BVHs (the top level follows them)
        uint32_t n_mesh_bvhs = 0, n_mesh_nodes = 0, n_mesh_prims = 0, n_faces = 0, n_vertices = 0;
        uint64_t hash = 0;                          // of the mesh arrays of the rt_scene_desc
        bool donate_ok = true;
    } mesh;

    // work memory
    Pool pool{};
    DevBuf b_ray0, b_ray1, b_hit0, b_hit1, b_nrm, b_link, b_color, b_shadow_rec, b_shadow_lit;
    DevBuf b_ctl, b_spill, b_cand, b_mt, b_frames, b_pstate, b_tile_order, b_fb, b_hits;
    DevBuf b_wq, b_sq;          // rt_mega.h: work rings of the trace workgroups, rings to their shading waves
    Control* h_ctl = nullptr;   // pinned
    void* h_stage = nullptr;    // pinned staging for rt_render_tile's host outputs
    size_t h_stage_bytes = 0;

    // options
    uint64_t opt_flags = 0;
    uint64_t opt_pool_records = 256ull << 20;    // upper bound; a render allocates what it needs (ensure_pool)
    uint64_t opt_chunk_samples = 64ull << 20;

    // timing events
    struct Ev { hipEvent_t a, b; int cat; };
    std::vector<Ev> ev_used, ev_free;
    hipEvent_t ev_total_a = nullptr, ev_total_b = nullptr;   // rt_stats.ms_total (created at upload, reused)

    // A render that returned before its kernels ran (rt_render_tile_device without stats) leaves its Control block
    // in h_ctl behind ev_async; rt_sync and the next render / update call look at it (check_async).
    hipEvent_t ev_async = nullptr;
    hipEvent_t ev_gather = nullptr;   // rt_gather_rows: this GPU's bands have been copied into GPU 0's frame
    bool pending_check = false;
    bool captured_check = false;      // a render was recorded into a HIP graph: its replays leave their Control block in h_ctl

    // deepest traversal stack any ray of this scene can need (entries), from the BVH depths found at upload
    uint32_t mesh_stack_need = 0, stack_cap = RT_STACK_MIN_CAP;
    uint32_t stack_cap_derived = RT_STACK_MIN_CAP;   // what the last upload / update derived from the BVH depths
    uint32_t stack_cap_override = 0;   // RT_OPT_STACK_CAP (diagnostic)
    unsigned char* static_ws = nullptr;   // this GPU's copy of `scratch` (device address)
};

namespace {

// ---------------------------------------------------------------- validation
int validate_bvh(const rt_bvh_node* nodes, uint32_t count, uint32_t n_prims, const char* what) {
    if (count == 0) return RT_OK;
    // The node array must be a TREE over all `count` nodes, as the reference's builders produce it (root 0,
    // children allocated pairwise after their parent, nodesUsed counts exactly the nodes reachable:
    // raytracer.cpp:11,189-190,323-324): node 0 has no parent, every other node exactly one.  The upload
    // relies on it — the device-side collapse (rt_upload.h) reads every node's parent link and reserves
    // count/2 + 1 records, and a shared child (a DAG) would make the host walks exponential.
    std::vector<uint8_t> parents(count, 0);
    for (uint32_t i = 0; i < count; i++) {
        const rt_bvh_node& n = nodes[i];
        if (n.prim_count > 0) {
            if (n.left_or_first < 0 || (uint64_t)n.left_or_first + (uint64_t)n.prim_count > n_prims)
                return fail(RT_ERR_INVALID, "%s: leaf %u range out of bounds", what, i);
        } else if (n.prim_count == 0) {
            // children are allocated after their parent: enforcing it guarantees the device traversal terminates
            if (n.left_or_first <= (int32_t)i || (uint32_t)n.left_or_first + 1 >= count)
                return fail(RT_ERR_INVALID, "%s: interior node %u has bad children", what, i);
            for (uint32_t c = (uint32_t)n.left_or_first; c <= (uint32_t)n.left_or_first + 1; c++)
                if (++parents[c] > 1) return fail(RT_ERR_INVALID, "%s: node %u has more than one parent", what, c);
        } else {
            return fail(RT_ERR_INVALID, "%s: node %u has negative prim_count", what, i);
        }
    }
    for (uint32_t i = 1; i < count; i++)
        if (parents[i] != 1) return fail(RT_ERR_INVALID, "%s: node %u is not referenced by any interior node", what, i);
    return RT_OK;
}

// Depth of a validated BVH (edges from the root to its deepest node).  Children follow their parent in the
// array, so one forward pass does it.
uint32_t bvh_depth(const rt_bvh_node* nodes, uint32_t count) {
    std::vector<uint32_t> depth(count, 0);
    uint32_t deepest = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (nodes[i].prim_count != 0) continue;
        const uint32_t c = (uint32_t)nodes[i].left_or_first, d = depth[i] + 1;
        depth[c] = depth[c + 1] = d;
        deepest = std::max(deepest, d);
    }
    return deepest;
}
// Stack entries the traversal of rt_traverse.h can hold at once for a BVH of that depth: a node record covers two
// reference levels and leaves up to three pending siblings behind.  (The reference's own walk keeps one sibling
// per level, at most depth + 1 of its stack[64], raytracer.cpp:533,766.)
uint32_t stack_need_of_depth(uint32_t depth) { return 3u * ((depth + 1u) / 2u);
}

int validate_scene(const rt_scene_desc* s, bool mesh_geometry = true) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    if (s->n_materials == 0) return fail(RT_ERR_INVALID, "scene has no materials");
    auto mat_ok = [&](int32_t id) { return id >= 1 && (uint32_t)id <= s->n_materials; };
    auto vtx_ok = [&](int32_t id) { return id >= 1 && (uint32_t)id <= s->n_vertices; };
    int rc = validate_bvh(s->top_nodes, s->n_top_nodes, s->n_top_prims, "top BVH");
    if (rc) return rc;
    for (uint32_t i = 0; i < s->n_top_prims; i++) {
        if (s->top_prim_idx[i] >= s->n_top_prims) return fail(RT_ERR_INVALID, "top_prim_idx[%u] out of range", i);
        const rt_top_prim& p = s->top_prims[i];
        uint32_t lim = p.kind == RT_KIND_MESH ? s->n_meshes : p.kind == RT_KIND_SPHERE ? s->n_spheres
                     : p.kind == RT_KIND_TRIANGLE ? s->n_triangles : 0;
        if (p.index >= lim) return fail(RT_ERR_INVALID, "top prim %u: bad kind/index", i);
    }
    for (uint32_t b = 0; mesh_geometry && b < s->n_mesh_bvhs; b++) {
        const rt_mesh_bvh& mb = s->mesh_bvhs[b];
        if ((uint64_t)mb.node_offset + mb.node_count > s->n_mesh_nodes || (uint64_t)mb.prim_offset + mb.prim_count > s->n_mesh_prims ||
            (uint64_t)mb.face_offset + mb.face_count > s->n_faces)
            return fail(RT_ERR_INVALID, "mesh BVH %u: offsets out of range", b);
        rc = validate_bvh(s->mesh_nodes + mb.node_offset, mb.node_count, mb.prim_count, "mesh BVH");
        if (rc) return rc;
        for (uint32_t p = 0; p < mb.prim_count; p++)
            if (s->mesh_prim_face[mb.prim_offset + p] >= mb.face_count) return fail(RT_ERR_INVALID, "mesh BVH %u: face index out of range", b);
        for (uint32_t f = 0; f < mb.face_count; f++) {
            const rt_face& fc = s->faces[mb.face_offset + f];
            if (!vtx_ok(fc.i0) || !vtx_ok(fc.i1) || !vtx_ok(fc.i2)) return fail(RT_ERR_INVALID, "face %u: vertex id out of range", f);
        }
    }
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        const rt_mesh& m = s->meshes[i];
        if (m.bvh_index < 0 || (uint32_t)m.bvh_index >= s->n_mesh_bvhs) return fail(RT_ERR_INVALID, "mesh %u: bad bvh_index", i);
        if (!mat_ok(m.material_id)) return fail(RT_ERR_INVALID, "mesh %u: bad material id", i);
    }
    for (uint32_t i = 0; i < s->n_triangles; i++) {
        const rt_triangle& t = s->triangles[i];
        if (!mat_ok(t.material_id) || !vtx_ok(t.face.i0) || !vtx_ok(t.face.i1) || !vtx_ok(t.face.i2))
            return fail(RT_ERR_INVALID, "triangle %u: bad ids", i);
    }
    for (uint32_t i = 0; i < s->n_spheres; i++)
        if (!mat_ok(s->spheres[i].material_id) || !vtx_ok(s->spheres[i].center_vertex_id)) return fail(RT_ERR_INVALID, "sphere %u: bad ids", i);
    for (uint32_t i = 0; i < s->n_planes; i++)
        if (!mat_ok(s->planes[i].material_id)) return fail(RT_ERR_INVALID, "plane %u: bad material id", i);
    return RT_OK;
}

// Mat4f::determinant3x3, Mat4f.h:126-130
float mat_det3(const float* m) {
    return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
}

// Re-layout of one reference BVH into DNode4 records (two reference levels per
// record, slots in the reference's depth-first order); leaves become links into
// the leaf-ordered prim array, with their prim count in the link.
struct BvhLayout {
    uint32_t root_link = RT_LINK_DONE;
    float rmin[3] = {0, 0, 0}, rmax[3] = {0, 0, 0};
};
struct BvhBuilder4 {
    const rt_bvh_node* nodes;
    uint32_t leaf_base;
    bool top;
    std::vector<DNode4>& out;
    std::vector<DBigLeaf>& big;
    uint32_t base;          // record index of out[0] in the device array

    uint32_t leaf_link(const rt_bvh_node& n) {
        const uint32_t first = leaf_base + (uint32_t)n.left_or_first;
        uint32_t code = (uint32_t)n.prim_count;
        if (code > 30) {
            big.push_back(DBigLeaf{first | (top ? 0x80000000u : 0u), (uint32_t)n.prim_count});
            code = RT_LEAF_BIG;
        }
        return RT_LINK_LEAF | (code << 26) | first;
    }
    // iterative (an unsplittable-heavy tree can be deep): work items are reference interior nodes
    uint32_t build(uint32_t root) {
        struct Item { uint32_t ref_node, slot; };
        std::vector<Item> work;
        const uint32_t root_slot = base + (uint32_t)out.size();
        out.emplace_back();
        work.push_back({root, root_slot});
        while (!work.empty()) {
            Item it = work.back();
            work.pop_back();
            DNode4 d;
            std::memset(&d, 0, sizeof(d));
            for (int k = 0; k < 4; k++) d.link[k] = RT_LINK_DONE;
            int s = 0;
            auto put = [&](int slot, uint32_t ref) {
                const rt_bvh_node& c = nodes[ref];
                float* f = d.f + slot;          // planes by axis (rt_device.h)
                f[0] = c.bmin[0]; f[4] = c.bmax[0]; f[8] = c.bmin[1]; f[12] = c.bmax[1]; f[16] = c.bmin[2]; f[20] = c.bmax[2];
                d.mask |= 1u << slot;
                if (c.prim_count > 0) {
                    d.link[slot] = leaf_link(c);
                } else {
                    const uint32_t ns = base + (uint32_t)out.size();
                    out.emplace_back();
                    d.link[slot] = ns;
                    work.push_back({ref, ns});
                }
            };
            const uint32_t L = (uint32_t)nodes[it.ref_node].left_or_first;
            for (uint32_t child : {L, L + 1}) {
                const rt_bvh_node& c = nodes[child];
                if (c.prim_count > 0) {
                    put(s, child);          // a leaf child keeps one slot; its sibling slot stays empty
                } else {
                    put(s, (uint32_t)c.left_or_first);
                    put(s + 1, (uint32_t)c.left_or_first + 1);
                }
                s += 2;
            }
            // slot 0 must be valid for the device's fast path; it is, since the left child always fills slot 0
            out[it.slot - base] = d;
        }
        return root_slot;
    }
};
BvhLayout layout_bvh(const rt_bvh_node* nodes, uint32_t count, uint32_t leaf_base, bool top, std::vector<DNode4>& out,
                     std::vector<DBigLeaf>& big, uint32_t base = 0) {
    BvhLayout L;
    if (count == 0) return L;
    out.reserve(out.size() + count / 2 + 16);       // one record per interior node of every other level
    BvhBuilder4 b{nodes, leaf_base, top, out, big, base};
    L.root_link = nodes[0].prim_count > 0 ? b.leaf_link(nodes[0]) : b.build(0);
    std::memcpy(L.rmin, nodes[0].bmin, 12);
    std::memcpy(L.rmax, nodes[0].bmax, 12);
    return L;
}

// Subtree donation (rt_traverse.h) orders the hits of one ray by triangle slot, which is the visiting
// order only if the depth-first, left-first walk meets the leaves with ascending first-prim index — true
// for the reference's in-place partitioning builder (Bvh.h / raytracer.cpp:247-262), checked here for any caller.
bool leaves_ascending(const rt_bvh_node* nodes, uint32_t count) {
    if (count == 0) return true;
    std::vector<uint32_t> stack{0};
    int64_t next = -1;
    while (!stack.empty()) {
        const rt_bvh_node& n = nodes[stack.back()];
        stack.pop_back();
        if (n.prim_count > 0) {
            if ((int64_t)n.left_or_first <= next) return false;
            next = (int64_t)n.left_or_first + n.prim_count - 1;
        } else {
            stack.push_back((uint32_t)n.left_or_first + 1);
            stack.push_back((uint32_t)n.left_or_first);
        }
    }
    return true;
}

void fill_object_common(DObject& o, uint32_t flags, const float* M, const float* Minv, const float* mb, int material_id) {
    std::memset(&o, 0, sizeof(o));
    std::memcpy(o.M, M, 64);
    std::memcpy(o.Minv, Minv, 64);
    o.mbx = mb[0]; o.mby = mb[1]; o.mbz = mb[2];
    o.flags = flags & (RT_OBJ_HAS_TRANSFORM | RT_OBJ_HAS_MOTION_BLUR | RT_OBJ_SMOOTH);
    if ((flags & RT_OBJ_HAS_TRANSFORM) && mat_det3(M) < 0.0f) o.flags |= DOBJ_NEG_DET;
    o.material_id = material_id;
    o.root_link = RT_LINK_DONE;
}

// 64-bit hash of the mesh arrays of a scene description (what rt_update_scene must find unchanged)
uint64_t hash_bytes(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    // four independent multiply-xor chains over 32-byte blocks (one chain is latency-bound), folded at the end
    uint64_t h0 = h, h1 = h ^ 0x243F6A8885A308D3ull, h2 = h ^ 0x13198A2E03707344ull, h3 = h ^ 0xA4093822299F31D0ull;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, b + i, 32);
        h0 = (h0 ^ w[0]) * 0x9E3779B97F4A7C15ull; h0 ^= h0 >> 29;
        h1 = (h1 ^ w[1]) * 0x9E3779B97F4A7C15ull; h1 ^= h1 >> 29;
        h2 = (h2 ^ w[2]) * 0x9E3779B97F4A7C15ull; h2 ^= h2 >> 29;
        h3 = (h3 ^ w[3]) * 0x9E3779B97F4A7C15ull; h3 ^= h3 >> 29;
    }
    h = h0;
    h = (h ^ h1) * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    h = (h ^ h2) * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    h = (h ^ h3) * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    for (; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}
uint64_t mesh_hash(const rt_scene_desc* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    h = hash_bytes(h, s->mesh_bvhs, sizeof(rt_mesh_bvh) * s->n_mesh_bvhs);
    h = hash_bytes(h, s->mesh_nodes, sizeof(rt_bvh_node) * s->n_mesh_nodes);
    h = hash_bytes(h, s->mesh_prim_face, sizeof(uint32_t) * s->n_mesh_prims);
    h = hash_bytes(h, s->faces, sizeof(rt_face) * s->n_faces);
    h = hash_bytes(h, s->vertex_data, sizeof(rt_vertex) * s->n_vertices);
    return h;
}

// Upload-time re-layout on the device: the leaf-ordered triangle records and their shading twins are gathered
// from the reference's own arrays (primIdx -> prims[].faceIdx -> faces[] -> vertex_data[], raytracer.cpp:783-789),
// with the same FP32 subtractions the host would do (col_A0 = a - b, col_A1 = a - c, raytracer.cpp:902-903).
__global__ void __launch_bounds__(256) k_build_tris(const uint32_t* __restrict__ prim_face, const rt_face* __restrict__ faces,
                                                    const rt_vertex* __restrict__ vd, uint32_t prim_offset, uint32_t face_offset,
                                                    uint32_t n, DTri* __restrict__ tris, DTriShade* __restrict__ tshade) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t fi = prim_face[prim_offset + p];
    const rt_face f = faces[face_offset + fi];
    const float* a = vd[f.i0 - 1].pos;
    const float* b = vd[f.i1 - 1].pos;
    const float* c = vd[f.i2 - 1].pos;
    const float ax = a[0], ay = a[1], az = a[2];
    DTri t;
    t.ax = ax; t.ay = ay; t.az = az;
    t.e1x = ax - b[0]; t.e1y = ay - b[1]; t.e1z = az - b[2];
    t.e2x = ax - c[0]; t.e2y = ay - c[1]; t.e2z = az - c[2];
    tris[prim_offset + p] = t;
    DTriShade ts;
    ts.nx = f.n_unit[0]; ts.ny = f.n_unit[1]; ts.nz = f.n_unit[2];
    ts.i0 = f.i0; ts.i1 = f.i1; ts.i2 = f.i2;
    ts.face = fi;
    ts.pad = 0;
    tshade[prim_offset + p] = ts;
}
__global__ void __launch_bounds__(256) k_build_vnormals(const rt_vertex* __restrict__ vd, uint32_t n, float4* __restrict__ vn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vn[i] = make_float4(vd[i].normal[0], vd[i].normal[1], vd[i].normal[2], 0.f);
}

#ifndef RT_COLLAPSE_MIN_NODES
#define RT_COLLAPSE_MIN_NODES 8192   // mesh BVHs below this are collapsed on the host
#endif

// RT_UPLOAD_TIMING=1: phase times of rt_upload_scene on stderr
struct PhaseClock {
    bool on = getenv("RT_UPLOAD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[rt upload] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// The mesh level: collapsed mesh BVHs, leaf-ordered triangle records, shading data, vertex normals.
// (hash_of: the description whose mesh arrays rt_update_scene will be compared with — the caller's, also when the trees
// were rebuilt.  The hash and the two host walks over the node arrays run on threads of their own beside the device work.)
int build_mesh_level(rt_scene* sc, const rt_scene_desc* s, const rt_scene_desc* hash_of) {
    PhaseClock pc;
    std::vector<DNode4> nodes;
    std::vector<DBigLeaf> big;
    std::future<uint64_t> fut_hash = std::async(std::launch::async, [hash_of] { return mesh_hash(hash_of); });
    struct Walk { bool ascending; uint32_t stack_need; };
    std::future<Walk> fut_walk = std::async(std::launch::async, [s] {
        Walk w{true, 0u};
        for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
            const rt_bvh_node* nd = s->mesh_nodes + s->mesh_bvhs[b].node_offset;
            if (!leaves_ascending(nd, s->mesh_bvhs[b].node_count)) w.ascending = false;
            w.stack_need = std::max(w.stack_need, stack_need_of_depth(bvh_depth(nd, s->mesh_bvhs[b].node_count)));
        }
        return w;
    });

    // mesh BVHs -> node records.  Large BVHs are collapsed on the device (rt_upload.h), small ones on the host;
    // the records of BVH b follow those of BVH b - 1 in the node array either way.
    const bool host_only = getenv("RT_UPLOAD_HOST_COLLAPSE") != nullptr;
    std::vector<BvhLayout> mesh_layout(s->n_mesh_bvhs);
    bool donate_ok = true;
    size_t rec_cap = 0;
    uint32_t max_dev_nodes = 0;
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
        rec_cap += s->mesh_bvhs[b].node_count / 2 + 1;
        if (!host_only && s->mesh_bvhs[b].node_count >= RT_COLLAPSE_MIN_NODES) max_dev_nodes = std::max(max_dev_nodes, s->mesh_bvhs[b].node_count);
    }
    // (RT_NODE_RECORDS_MAX: the traversal addresses node records by 32-bit byte offsets, 128 B each — 4 GB of records;
    // with at most 2^26 leaf prims a scene has about 2^25 of them, so the prim limit is the one that is met first)
    if (rec_cap >= RT_NODE_RECORDS_MAX - 2ull * s->n_top_nodes || s->n_mesh_prims >= RT_LEAF_MAX_FIRST)
        return fail(RT_ERR_UNSUPPORTED, "scene too large for the link encoding (2^26 leaf prims, 2^25 node records)");
    int rc;
    // the top level's records follow the meshes' in the same array: leave room for them (rt_update_scene grows it if needed)
    if ((rc = sc->b_nodes.ensure((rec_cap + std::max<size_t>(1024, 2 * (size_t)s->n_top_nodes)) * sizeof(DNode4)))) return rc;
    DevBuf d_ref, d_anc0, d_anc1, d_local, d_sums, d_word;
    struct Release6 { DevBuf *b[6]; ~Release6() { for (DevBuf* x : b) x->release(); } } rel6{{&d_ref, &d_anc0, &d_anc1, &d_local, &d_sums, &d_word}};
    if (max_dev_nodes) {
        const size_t nb = (max_dev_nodes + RT_COLLAPSE_BLOCK - 1) / RT_COLLAPSE_BLOCK;
        if ((rc = d_ref.ensure((size_t)s->n_mesh_nodes * sizeof(rt_bvh_node))) || (rc = d_anc0.ensure((size_t)max_dev_nodes * 4)) ||
            (rc = d_anc1.ensure((size_t)max_dev_nodes * 4)) || (rc = d_local.ensure((size_t)max_dev_nodes * 4)) ||
            (rc = d_sums.ensure(nb * 4)) || (rc = d_word.ensure(8)))
            return rc;
        HIPCHK(hipMemcpy(d_ref.p, s->mesh_nodes, (size_t)s->n_mesh_nodes * sizeof(rt_bvh_node), hipMemcpyHostToDevice));
    }
    uint32_t rec_count = 0;                     // records written (or pending in `nodes`) so far
    uint32_t pending_base = 0;                  // device index of nodes[0]
    auto flush_host_records = [&]() -> int {
        if (!nodes.empty()) HIPCHK(hipMemcpy((DNode4*)sc->b_nodes.p + pending_base, nodes.data(), nodes.size() * sizeof(DNode4), hipMemcpyHostToDevice));
        nodes.clear();
        pending_base = rec_count;
        return RT_OK;
    };
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
        const rt_mesh_bvh& mb = s->mesh_bvhs[b];
        const rt_bvh_node* ref = s->mesh_nodes + mb.node_offset;
        if (host_only || mb.node_count < RT_COLLAPSE_MIN_NODES) {
            mesh_layout[b] = layout_bvh(ref, mb.node_count, mb.prim_offset, false, nodes, big, pending_base);
            rec_count = pending_base + (uint32_t)nodes.size();
            continue;
        }
        if ((rc = flush_host_records())) return rc;
        const uint32_t n = mb.node_count, grid = (n + 255) / 256, nb = (n + RT_COLLAPSE_BLOCK - 1) / RT_COLLAPSE_BLOCK;
        const rt_bvh_node* dref = (const rt_bvh_node*)d_ref.p + mb.node_offset;
        uint32_t* anc[2] = {(uint32_t*)d_anc0.p, (uint32_t*)d_anc1.p};
        uint32_t* word = (uint32_t*)d_word.p;
        k_collapse_parents<<<grid, 256, 0, sc->stream>>>(dref, n, anc[0]);
        int cur = 0;
        bool converged = false;
        for (int round = 0; round < 12 && !converged; round++) {      // depth <= 4096; the reference's own stack holds 64
            const int jumps = round == 0 ? 6 : 1;
            for (int j = 0; j < jumps; j++) {
                HIPCHK(hipMemsetAsync(word, 0, 4, sc->stream));
                k_collapse_jump<<<grid, 256, 0, sc->stream>>>(anc[cur], anc[cur ^ 1], n, word);
                cur ^= 1;
            }
            uint32_t not_done = 1;
            HIPCHK(hipMemcpyAsync(&not_done, word, 4, hipMemcpyDeviceToHost, sc->stream));
            HIPCHK(hipStreamSynchronize(sc->stream));
            converged = not_done == 0;
        }
        if (!converged) return fail(RT_ERR_INVALID, "mesh BVH %u is deeper than 4096 levels", b);
        k_collapse_flag_scan<<<nb, RT_COLLAPSE_BLOCK, 0, sc->stream>>>(dref, anc[cur], n, (uint32_t*)d_local.p, (uint32_t*)d_sums.p);
        k_collapse_scan_sums<<<1, RT_COLLAPSE_BLOCK, 0, sc->stream>>>((uint32_t*)d_sums.p, nb, word + 1);
        uint32_t n_rec = 0;
        HIPCHK(hipMemcpyAsync(&n_rec, word + 1, 4, hipMemcpyDeviceToHost, sc->stream));
        HIPCHK(hipStreamSynchronize(sc->stream));
        // a tree of n nodes has at most n / 2 interior nodes, hence record roots (validate_bvh established the
        // tree); never let the emit kernel write beyond what this BVH reserved in the node array
        if (n_rec > n / 2 + 1) return fail(RT_ERR_INVALID, "mesh BVH %u: %u node records for %u nodes (not a tree)", b, n_rec, n);
        k_collapse_emit<<<grid, 256, 0, sc->stream>>>(dref, (const uint32_t*)d_local.p, (const uint32_t*)d_sums.p, n, mb.prim_offset, rec_count,
                                                      (DNode4*)sc->b_nodes.p);
        HIPCHK(hipStreamSynchronize(sc->stream));
        HIPCHK(hipGetLastError());
        BvhLayout& L = mesh_layout[b];
        std::memcpy(L.rmin, ref[0].bmin, 12);
        std::memcpy(L.rmax, ref[0].bmax, 12);
        for (uint32_t i = 0; i < n; i++)            // leaves beyond the 5-bit count: side table (sorted later)
            if (ref[i].prim_count > 30) big.push_back(DBigLeaf{mb.prim_offset + (uint32_t)ref[i].left_or_first, (uint32_t)ref[i].prim_count});
        if (ref[0].prim_count > 0) {
            BvhBuilder4 tmp{ref, mb.prim_offset, false, nodes, big, 0};
            const size_t nbig = big.size();
            L.root_link = tmp.leaf_link(ref[0]);
            big.resize(nbig);                       // (already listed by the loop above)
        } else {
            L.root_link = rec_count;
        }
        rec_count += n_rec;
        pending_base = rec_count;
    }
    if ((rc = flush_host_records())) return rc;
    pc.lap("mesh BVH collapse");
    // triangle records, shading records and vertex normals: gathered on the device from the caller's arrays
    if ((rc = sc->b_tris.ensure(std::max<size_t>(s->n_mesh_prims, 1) * sizeof(DTri))) ||
        (rc = sc->b_tri_shade.ensure(std::max<size_t>(s->n_mesh_prims, 1) * sizeof(DTriShade))) ||
        (rc = sc->b_vnormals.ensure(std::max<size_t>(s->n_vertices, 1) * sizeof(float4))))
        return rc;
    {
        DevBuf d_prim_face, d_faces, d_vd;
        struct Release { DevBuf &a, &b, &c; ~Release() { a.release(); b.release(); c.release(); } } rel{d_prim_face, d_faces, d_vd};
        if ((rc = d_prim_face.ensure(std::max<size_t>(s->n_mesh_prims, 1) * sizeof(uint32_t))) ||
            (rc = d_faces.ensure(std::max<size_t>(s->n_faces, 1) * sizeof(rt_face))) ||
            (rc = d_vd.ensure(std::max<size_t>(s->n_vertices, 1) * sizeof(rt_vertex))))
            return rc;
        if (s->n_mesh_prims) HIPCHK(hipMemcpy(d_prim_face.p, s->mesh_prim_face, (size_t)s->n_mesh_prims * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (s->n_faces) HIPCHK(hipMemcpy(d_faces.p, s->faces, (size_t)s->n_faces * sizeof(rt_face), hipMemcpyHostToDevice));
        if (s->n_vertices) HIPCHK(hipMemcpy(d_vd.p, s->vertex_data, (size_t)s->n_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice));
        for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
            const rt_mesh_bvh& mb = s->mesh_bvhs[b];
            if (mb.prim_count == 0) continue;
            k_build_tris<<<(mb.prim_count + 255) / 256, 256, 0, sc->stream>>>((const uint32_t*)d_prim_face.p, (const rt_face*)d_faces.p, (const rt_vertex*)d_vd.p,
                                                                           mb.prim_offset, mb.face_offset, mb.prim_count, (DTri*)sc->b_tris.p,
                                                                           (DTriShade*)sc->b_tri_shade.p);
        }
        if (s->n_vertices)
            k_build_vnormals<<<(s->n_vertices + 255) / 256, 256, 0, sc->stream>>>((const rt_vertex*)d_vd.p, s->n_vertices, (float4*)sc->b_vnormals.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(sc->stream));
    }
    pc.lap("copies + device re-layout");

    rt_scene::MeshLevel& M = sc->mesh;
    M.roots.resize(s->n_mesh_bvhs);
    for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
        M.roots[b].root_link = mesh_layout[b].root_link;
        std::memcpy(M.roots[b].rmin, mesh_layout[b].rmin, 12);
        std::memcpy(M.roots[b].rmax, mesh_layout[b].rmax, 12);
    }
    M.big = big;
    M.n_node_records = rec_count;
    M.n_mesh_bvhs = s->n_mesh_bvhs; M.n_mesh_nodes = s->n_mesh_nodes; M.n_mesh_prims = s->n_mesh_prims;
    M.n_faces = s->n_faces; M.n_vertices = s->n_vertices;
    {
        const Walk w = fut_walk.get();
        if (!w.ascending) donate_ok = false;
        sc->mesh_stack_need = w.stack_need;
    }
    M.hash = fut_hash.get();
    pc.lap("leaf order + depth, mesh hash (host threads, waited for)");
    M.donate_ok = donate_ok;
    sc->ds.tris = (const DTri*)sc->b_tris.p;
    sc->ds.tri_shade = (const DTriShade*)sc->b_tri_shade.p;
    sc->ds.vnormals = (const float4*)sc->b_vnormals.p;
    return RT_OK;
}

// Everything else: objects (transforms, motion blur, materials), the top-level BVH, planes, materials, lights,
// scene constants.  Needs the mesh level of `sc` (build_mesh_level) and leaves it untouched.
int build_top_level(rt_scene* sc, const rt_scene_desc* s) {
    const rt_scene::MeshLevel& M = sc->mesh;
    std::vector<DNode4> nodes;                   // the top level's records; record 0 of it is device record M.n_node_records
    std::vector<DBigLeaf> big = M.big;

    // objects: meshes | spheres | triangles
    std::vector<DObject> objects(s->n_meshes + s->n_spheres + s->n_triangles);
    sc->ds.mesh_obj_base = 0;
    sc->ds.sphere_obj_base = s->n_meshes;
    sc->ds.tri_obj_base = s->n_meshes + s->n_spheres;
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        const rt_mesh& m = s->meshes[i];
        DObject& o = objects[i];
        fill_object_common(o, m.flags, m.transform, m.inv_transform, m.motion_blur, m.material_id);
        const MeshRoot& L = M.roots[m.bvh_index];
        o.root_link = L.root_link;
        o.rminx = L.rmin[0]; o.rminy = L.rmin[1]; o.rminz = L.rmin[2];
        o.rmaxx = L.rmax[0]; o.rmaxy = L.rmax[1]; o.rmaxz = L.rmax[2];
    }
    for (uint32_t i = 0; i < s->n_spheres; i++) {
        const rt_sphere& sp = s->spheres[i];
        DObject& o = objects[sc->ds.sphere_obj_base + i];
        fill_object_common(o, sp.flags, sp.transform, sp.inv_transform, sp.motion_blur, sp.material_id);
        const float* c = s->vertex_data[sp.center_vertex_id - 1].pos;
        o.p[0] = c[0]; o.p[1] = c[1]; o.p[2] = c[2]; o.p[3] = sp.radius;
    }
    for (uint32_t i = 0; i < s->n_triangles; i++) {
        const rt_triangle& t = s->triangles[i];
        DObject& o = objects[sc->ds.tri_obj_base + i];
        fill_object_common(o, t.flags, t.transform, t.inv_transform, t.motion_blur, t.material_id);
        const float* a = s->vertex_data[t.face.i0 - 1].pos;
        const float* b = s->vertex_data[t.face.i1 - 1].pos;
        const float* c = s->vertex_data[t.face.i2 - 1].pos;
        o.p[0] = a[0]; o.p[1] = a[1]; o.p[2] = a[2];
        o.p[3] = a[0] - b[0]; o.p[4] = a[1] - b[1]; o.p[5] = a[2] - b[2];
        o.p[6] = a[0] - c[0]; o.p[7] = a[1] - c[1]; o.p[8] = a[2] - c[2];
        o.p[9] = t.face.n_unit[0]; o.p[10] = t.face.n_unit[1]; o.p[11] = t.face.n_unit[2];
    }

    // top level: slots in topPrimIdx order
    std::vector<DTopPrim> top(s->n_top_prims);
    for (uint32_t k = 0; k < s->n_top_prims; k++) {
        const rt_top_prim& p = s->top_prims[s->top_prim_idx[k]];
        top[k].kind = p.kind;
        top[k].index = p.index;
        top[k].remaining = 1;
        top[k].obj = p.kind == RT_KIND_MESH ? sc->ds.mesh_obj_base + p.index
                   : p.kind == RT_KIND_SPHERE ? sc->ds.sphere_obj_base + p.index : sc->ds.tri_obj_base + p.index;
    }
    for (uint32_t i = 0; i < s->n_top_nodes; i++) {
        const rt_bvh_node& n = s->top_nodes[i];
        for (int32_t k = 0; k < n.prim_count; k++) top[n.left_or_first + k].remaining = (uint32_t)(n.prim_count - k);
    }
    // the top level in the reference's visiting order: nodes in pre-order, a leaf's prims behind it (rt_device.h: RT_TOPITEM_*)
    std::vector<uint4> items;
    if (s->n_top_prims != 0) {
        items.reserve(2 * ((size_t)s->n_top_nodes + (size_t)s->n_top_prims * (1 + RT_INLINE_LEAF_MAX)));
        struct Visit { uint32_t node; uint32_t item; bool close; };
        std::vector<Visit> work{{0u, 0u, false}};
        while (!work.empty()) {
            const Visit v = work.back();
            work.pop_back();
            if (v.close) { items[2 * (size_t)v.item + 1].z = (uint32_t)(items.size() / 2); continue; }
            const rt_bvh_node& n = s->top_nodes[v.node];
            const uint32_t at = (uint32_t)(items.size() / 2);
            uint4 a, b;
            std::memcpy(&a.x, &n.bmin[0], 4); std::memcpy(&a.y, &n.bmin[1], 4); std::memcpy(&a.z, &n.bmin[2], 4); std::memcpy(&a.w, &n.bmax[0], 4);
            std::memcpy(&b.x, &n.bmax[1], 4); std::memcpy(&b.y, &n.bmax[2], 4);
            b.z = 0u; b.w = RT_TOPITEM_NODE;
            items.push_back(a); items.push_back(b);
            work.push_back({0u, at, true});                 // (runs after the subtree: the item that follows it)
            if (n.prim_count > 0) {
                for (int32_t k = 0; k < n.prim_count; k++) {
                    const DTopPrim& tp = top[n.left_or_first + k];
                    const DObject& o = objects[tp.obj];
                    const uint32_t leaf_n = RT_LEAF_COUNT(o.root_link);
                    if (RT_INLINE_LEAF_MESH && tp.kind == RT_KIND_MESH && !(o.flags & (RT_OBJ_HAS_TRANSFORM | RT_OBJ_HAS_MOTION_BLUR)) &&
                        (o.root_link & RT_LINK_LEAF) && o.root_link < RT_LINK_TOPSCAN && leaf_n >= 1 && leaf_n <= RT_INLINE_LEAF_MAX) {
                        // the instance's root box as a node of its own, its triangles behind it (rt_device.h: RT_INLINE_LEAF_MESH)
                        uint4 a, b;
                        std::memcpy(&a.x, &o.rminx, 4); std::memcpy(&a.y, &o.rminy, 4); std::memcpy(&a.z, &o.rminz, 4); std::memcpy(&a.w, &o.rmaxx, 4);
                        std::memcpy(&b.x, &o.rmaxy, 4); std::memcpy(&b.y, &o.rmaxz, 4);
                        b.z = (uint32_t)(items.size() / 2) + 1u + leaf_n; b.w = RT_TOPITEM_NODE;
                        items.push_back(a); items.push_back(b);
                        for (uint32_t j = 0; j < leaf_n; j++) {
                            items.push_back(make_uint4(RT_KIND_MESHTRI, tp.index, tp.obj, RT_LEAF_FIRST(o.root_link) + j));
                            items.push_back(make_uint4(0u, 0u, 0u, RT_TOPITEM_PRIM));
                        }
                        continue;
                    }
                    items.push_back(make_uint4(tp.kind, tp.index, tp.obj, 0u));
                    items.push_back(make_uint4(0u, 0u, 0u, RT_TOPITEM_PRIM));
                }
            } else {
                work.push_back({(uint32_t)n.left_or_first + 1u, 0u, false});
                work.push_back({(uint32_t)n.left_or_first, 0u, false});
            }
        }
    }
    if (items.size() / 2 >= 0x7FFFFFFFu) return fail(RT_ERR_UNSUPPORTED, "top level too large");
    BvhLayout TL = layout_bvh(s->top_nodes, s->n_top_prims ? s->n_top_nodes : 0, 0, true, nodes, big, M.n_node_records);
    // one stack per ray: the sentinel and the mesh level (and, without RT_FLAT_TOP, the top-level entries below them)
#if RT_FLAT_TOP
    // (the top level is walked without a stack, rt_traverse.h: only the sentinel and the mesh level are on it)
    const uint32_t stack_need = 1u + sc->mesh_stack_need + 2u;
#else
    const uint32_t stack_need = stack_need_of_depth(bvh_depth(s->top_nodes, s->n_top_prims ? s->n_top_nodes : 0)) + 1u + sc->mesh_stack_need + 2u;
#endif
    if (stack_need > RT_STACK_CAP_LIMIT)
        return fail(RT_ERR_UNSUPPORTED, "BVH too deep: a ray could need %u traversal stack entries (limit %d)", stack_need, RT_STACK_CAP_LIMIT);
    sc->stack_cap_derived = std::max<uint32_t>(RT_STACK_MIN_CAP, stack_need);
    sc->stack_cap = sc->stack_cap_override ? sc->stack_cap_override : sc->stack_cap_derived;
    if ((uint64_t)M.n_node_records + nodes.size() >= RT_NODE_RECORDS_MAX || top.size() >= RT_LEAF_MAX_FIRST)
        return fail(RT_ERR_UNSUPPORTED, "scene too large for the link encoding (2^26 leaf prims, 2^25 node records)");
    std::sort(big.begin(), big.end(), [](const DBigLeaf& a, const DBigLeaf& b) { return a.first < b.first; });
    if (big.empty()) big.push_back(DBigLeaf{0, 0});

    std::vector<DPlane> planes(s->n_planes);
    for (uint32_t i = 0; i < s->n_planes; i++) {
        planes[i].nx = s->planes[i].n_unit[0]; planes[i].ny = s->planes[i].n_unit[1]; planes[i].nz = s->planes[i].n_unit[2];
        planes[i].d = s->planes[i].plane_d;
        planes[i].material_id = s->planes[i].material_id;
    }
    std::vector<DMaterial> mats(s->n_materials);
    sc->has_reflective = false;
    sc->needs_sequential_rng = s->n_area_lights > 0 || getenv("RT_FORCE_ORDERED") != nullptr;   // (diagnostic: every scene through the depth-first pixel loop)
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const rt_material& m = s->materials[i];
        DMaterial& d = mats[i];
        d.type = m.type;
        std::memcpy(d.ka, m.ambient_refl, 12); std::memcpy(d.kd, m.diffuse_refl, 12);
        std::memcpy(d.ks, m.specular_refl, 12); std::memcpy(d.km, m.mirror_refl, 12);
        d.phong = m.phong_exponent; d.ior = m.refraction_index; d.absorb_index = m.absorption_index;
        std::memcpy(d.absorb_coef, m.absorption_coef, 12);
        d.roughness = m.roughness;
        if (m.type != RT_MAT_NONE) {
            sc->has_reflective = true;
            if (m.roughness > 0.0f) sc->needs_sequential_rng = true;   // PerturbReflection draws (raytracer.cpp:1441-1452)
        }
    }
    std::vector<DPointLight> lights(s->n_point_lights);
    for (uint32_t i = 0; i < s->n_point_lights; i++) {
        lights[i].px = s->point_lights[i].position[0]; lights[i].py = s->point_lights[i].position[1]; lights[i].pz = s->point_lights[i].position[2];
        lights[i].ix = s->point_lights[i].intensity[0]; lights[i].iy = s->point_lights[i].intensity[1]; lights[i].iz = s->point_lights[i].intensity[2];
    }

    std::vector<DAreaLight> alights(s->n_area_lights);
    for (uint32_t i = 0; i < s->n_area_lights; i++) {
        const rt_area_light& a = s->area_lights[i];
        alights[i] = DAreaLight{a.position[0], a.position[1], a.position[2], a.normal[0], a.normal[1], a.normal[2], a.size,
                                a.radiance[0], a.radiance[1], a.radiance[2]};
    }
    sc->n_area_lights = s->n_area_lights;

    int rc;
    if ((rc = upload_vec(sc->b_area_lights, alights))) return rc;
    if ((rc = sc->b_nodes.grow_keep(((size_t)M.n_node_records + nodes.size()) * sizeof(DNode4), (size_t)M.n_node_records * sizeof(DNode4)))) return rc;
    if (!nodes.empty())
        HIPCHK(hipMemcpy((DNode4*)sc->b_nodes.p + M.n_node_records, nodes.data(), nodes.size() * sizeof(DNode4), hipMemcpyHostToDevice));
    if ((rc = upload_vec(sc->b_objects, objects)) || (rc = upload_vec(sc->b_top_prims, top)) || (rc = upload_vec(sc->b_top_items, items)) ||
        (rc = upload_vec(sc->b_big, big)) || (rc = upload_vec(sc->b_planes, planes)) || (rc = upload_vec(sc->b_materials, mats)) || (rc = upload_vec(sc->b_lights, lights)))
        return rc;

    DScene& D = sc->ds;
    D.nodes = (const DNode4*)sc->b_nodes.p;
    D.big_leaves = (const DBigLeaf*)sc->b_big.p;
    D.n_big_leaves = (uint32_t)big.size();
    D.stack_cap = sc->stack_cap;
    D.donate_ok = M.donate_ok ? 1u : 0u;
    if (getenv("RT_NO_DONATE")) D.donate_ok = 0u;      // diagnostic: every ray is traced by its owner alone
    D.objects = (const DObject*)sc->b_objects.p;
    D.top_prims = (const DTopPrim*)sc->b_top_prims.p;
    D.top_items = (const uint4*)sc->b_top_items.p;
    D.n_top_items = (uint32_t)(items.size() / 2);
    D.planes = (const DPlane*)sc->b_planes.p;
    D.materials = (const DMaterial*)sc->b_materials.p;
    D.point_lights = (const DPointLight*)sc->b_lights.p;
    D.n_planes = s->n_planes;
    D.n_point_lights = s->n_point_lights;
    D.n_materials = s->n_materials;
    D.top_root_link = TL.root_link;
    std::memcpy(D.top_rmin, TL.rmin, 12);
    std::memcpy(D.top_rmax, TL.rmax, 12);
    std::memcpy(D.background, s->background_color, 12);
    std::memcpy(D.ambient, s->ambient_light, 12);
    D.shadow_eps = s->shadow_ray_epsilon;
    D.max_depth = s->max_recursion_depth;
    if (sc->n_point_lights != s->n_point_lights) sc->pool.cap_records = 0;   // the shadow pool is sized by the light count
    sc->n_point_lights = s->n_point_lights;
    return RT_OK;
}

int build_device_scene(rt_scene* sc, const rt_scene_desc* s, uint32_t upload_flags) {
    int rc;
    if (upload_flags & RT_UPLOAD_REBUILD_MESH_BVH) {
        // non-parity mode: the mesh level is laid out from binned-SAH trees over the same faces (rt_rebuild.h);
        // everything downstream is the code of the parity path
        PhaseClock pc;
        uint32_t max_leaf = 4;
        if (const char* e = getenv("RT_REBUILD_MAX_LEAF")) max_leaf = (uint32_t)std::max(2, std::min(30, atoi(e)));
        rt_rebuild::MeshLevel R;
        if (upload_flags & RT_UPLOAD_REBUILD_ON_DEVICE) {
            const char* err = nullptr;
            if (upload_flags & RT_UPLOAD_REBUILD_LBVH) {
                if (!rt_lbvh::rebuild(s, max_leaf, R, err)) return fail(RT_ERR_HIP, "LBVH rebuild: %s", err ? err : "?");
                pc.lap("LBVH rebuild (device)");
            } else {
                if (!rt_dsah::rebuild(s, max_leaf, R, err)) return fail(RT_ERR_HIP, "SAH rebuild on the device: %s", err ? err : "?");
                pc.lap("SAH rebuild (device)");
            }
        } else {
            rt_rebuild::rebuild(s, max_leaf, R);
            pc.lap("SAH rebuild (host)");
        }
        if (getenv("RT_UPLOAD_TIMING")) {
            // diagnostic: the SAH cost of every rebuilt tree (node areas + leaf areas x triangles, over the root's area), so
            // that the builders can be compared without rendering — the device SAH builder must print the host builder's figures
            for (uint32_t b = 0; b < s->n_mesh_bvhs; b++) {
                const rt_bvh_node* nd = R.nodes.data() + R.bvhs[b].node_offset;
                auto area = [](const rt_bvh_node& n) { const double dx = (double)n.bmax[0] - n.bmin[0], dy = (double)n.bmax[1] - n.bmin[1], dz = (double)n.bmax[2] - n.bmin[2]; return dx * dy + dy * dz + dz * dx; };
                double cost = 0; uint64_t leaves = 0, tris = 0; uint32_t big = 0;
                for (uint32_t i = 0; i < R.bvhs[b].node_count; i++) {
                    if (nd[i].prim_count > 0) { cost += area(nd[i]) * nd[i].prim_count; leaves++; tris += (uint32_t)nd[i].prim_count; big = std::max(big, (uint32_t)nd[i].prim_count); }
                    else cost += area(nd[i]);
                }
                const double ra = R.bvhs[b].node_count ? area(nd[0]) : 0.0;
                fprintf(stderr, "[rt upload]   rebuilt BVH %u: %u nodes, %llu leaves (largest %u), %llu triangles, SAH cost %.6f\n", b, R.bvhs[b].node_count,
                        (unsigned long long)leaves, big, (unsigned long long)tris, ra > 0 ? cost / ra : 0.0);
            }
        }
        rt_scene_desc s2 = *s;
        s2.mesh_nodes = R.nodes.data(); s2.n_mesh_nodes = (uint32_t)R.nodes.size();
        s2.mesh_bvhs = R.bvhs.data();
        s2.mesh_prim_face = R.prim_face.data();
        for (uint32_t b = 0; b < s2.n_mesh_bvhs; b++)
            if ((rc = validate_bvh(s2.mesh_nodes + s2.mesh_bvhs[b].node_offset, s2.mesh_bvhs[b].node_count, s2.mesh_bvhs[b].prim_count, "rebuilt mesh BVH"))) return rc;
        if ((rc = build_mesh_level(sc, &s2, s))) return rc;
        // rt_update_scene compares the caller's arrays with what the caller uploaded (the hash is of `s` already)
        sc->mesh.n_mesh_nodes = s->n_mesh_nodes;
        if ((rc = upload_vec(sc->b_tri_rank, R.rank))) return rc;
        sc->ds.tri_rank = R.rank.empty() ? nullptr : (const uint32_t*)sc->b_tri_rank.p;
        sc->rebuilt = true;
    } else {
        if ((rc = build_mesh_level(sc, s, s))) return rc;
        sc->ds.tri_rank = nullptr;
    }
    return build_top_level(sc, s);
}

// The record pool grows to what a render needs, up to RT_OPT_POOL_RECORDS (27 GB at its default of 256 M records:
// sized for 288 GB of HBM — large pools mean few chunks, hence few launches with an end game each).
int ensure_pool(rt_scene* sc, uint64_t want_records, uint64_t slack_records = 0, uint64_t slack_shadow = 0) {
    // (slack: the slabs the shading waves of the one-launch pipeline hold but may not fill, rt_mega.h)
    const size_t cap = (size_t)std::min<uint64_t>(sc->opt_pool_records, std::max<uint64_t>(want_records, 1ull << 16) + slack_records);
    const size_t lights = std::max<uint32_t>(sc->n_point_lights, 1);
    const size_t cap_sh = cap * std::min<size_t>(lights, 4) + (size_t)slack_shadow;
    // (the shadow slots too: a pool sized by an earlier per-bounce render has no room for the one-launch pipeline's slabs)
    if (sc->pool.cap_records >= cap && sc->pool.cap_records <= sc->opt_pool_records && sc->pool.cap_shadow >= cap_sh && sc->b_ray0.p) return RT_OK;
    int rc;
    if ((rc = sc->b_ray0.ensure(cap * 16)) || (rc = sc->b_ray1.ensure(cap * 16)) || (rc = sc->b_hit0.ensure(cap * 16)) ||
        (rc = sc->b_hit1.ensure(cap * 8)) || (rc = sc->b_nrm.ensure(cap * 16)) || (rc = sc->b_link.ensure(cap * 16)) ||
        (rc = sc->b_color.ensure(cap * 16)) || (rc = sc->b_shadow_rec.ensure(cap_sh * 4)) || (rc = sc->b_shadow_lit.ensure(cap_sh)))
        return rc;
    Pool& P = sc->pool;
    P.ray0 = (float4*)sc->b_ray0.p; P.ray1 = (float4*)sc->b_ray1.p; P.hit0 = (uint4*)sc->b_hit0.p; P.hit1 = (float2*)sc->b_hit1.p;
    P.nrm = (float4*)sc->b_nrm.p; P.link = (uint4*)sc->b_link.p; P.color = (float4*)sc->b_color.p;
    P.shadow_rec = (uint32_t*)sc->b_shadow_rec.p; P.shadow_lit = (uint8_t*)sc->b_shadow_lit.p;
    P.cap_records = (uint32_t)cap;
    P.cap_shadow = (uint32_t)cap_sh;
    return RT_OK;
}

struct Timer {
    rt_scene* sc;
    hipStream_t st;
    bool on;
    int begin(int cat) {
        if (!on) return RT_OK;
        rt_scene::Ev e;
        if (!sc->ev_free.empty()) { e = sc->ev_free.back(); sc->ev_free.pop_back(); }
        else { HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); }
        e.cat = cat;
        HIPCHK(hipEventRecord(e.a, st));
        sc->ev_used.push_back(e);
        return RT_OK;
    }
    int end() {
        if (!on) return RT_OK;
        HIPCHK(hipEventRecord(sc->ev_used.back().b, st));
        return RT_OK;
    }
};

// Control::overflow as left by the kernels of a render -> status.  Bits 1 / 2 (record / shadow pool) are handled by
// the chunk loop where child rays exist; seen here they mean the pools were too small for what was asked.
int ctl_status(uint32_t ov, uint32_t stack_cap) {
    if (ov & 8u) return fail(RT_ERR_HIP, "internal: a traced ray reached the wrong lane (rt_ordered.h)");
    // (a full ring first: the workgroup it stops may also trip the bounds below)
    if (ov & 64u) return fail(RT_ERR_OVERFLOW, "work ring of a trace workgroup overflowed (rt_mega.h): render smaller tiles or set RT_FLAG_PER_BOUNCE_LAUNCHES");
    if (ov & 32u) return fail(RT_ERR_HIP, "internal: a wave exceeded its round bound (rt_traverse.h)");
    if (ov & 16u) return fail(RT_ERR_HIP, "internal: a ray waited for its helpers beyond the bound (rt_traverse.h)");
    if (ov & 4u) return fail(RT_ERR_OVERFLOW, "traversal stack deeper than %u entries", stack_cap);
    if (ov & 3u) return fail(RT_ERR_OVERFLOW, "ray pool overflow");
    return RT_OK;
}

// The Control block of the last render that returned before its kernels had run (rt.h: rt_render_tile_device /
// _rows_device without stats).  Called by rt_sync and at the start of the next render / update of the scene:
// waits for that render and reports what it left — once; the frame it wrote is not to be trusted then.
int check_async(rt_scene* sc) {
    if (sc->captured_check && !sc->pending_check) {
        // (no event to wait for: the caller has synchronised the stream it launched the graph on — rt.h)
        const int rc = ctl_status(sc->h_ctl->overflow, sc->stack_cap);
        if (rc) { const std::string msg = g_err; return fail(rc, "a render of this scene replayed from a HIP graph failed: %s", msg.c_str()); }
        return RT_OK;
    }
    if (!sc->pending_check) return RT_OK;
    sc->pending_check = false;
    HIPCHK(hipEventSynchronize(sc->ev_async));
    const int rc = ctl_status(sc->h_ctl->overflow, sc->stack_cap);
    if (rc) {
        const std::string msg = g_err;
        return fail(rc, "an earlier asynchronous render of this scene failed: %s", msg.c_str());
    }
    return RT_OK;
}

void to_dcamera(const rt_camera_desc* c, DCamera& d) {
    std::memcpy(d.pos, c->position, 12); std::memcpy(d.u, c->u, 12); std::memcpy(d.v, c->v, 12);
    std::memcpy(d.w, c->w, 12); std::memcpy(d.q, c->q, 12);
    d.pixel_w = c->pixel_width; d.pixel_h = c->pixel_height;
    d.image_w = c->image_width; d.image_h = c->image_height;
    d.num_samples = c->num_samples; d.sps = c->samples_per_side;
    d.aperture = c->aperture_size; d.focus = c->focus_distance;
    d.dof = c->has_depth_of_field;
}

// ws != nullptr: the no-allocation variant — all work memory is carved out of that GPU's static array instead of the
// scene's buffers, and nothing here may allocate (device or pinned memory, events).
int render_impl(rt_scene* sc, const rt_camera_desc* cam, int x0, int y0, int w, int h, int row_parts, int row_part,
                uint8_t* d_rgb, size_t stride, uint32_t* d_hits, rt_stats* stats, hipStream_t st, unsigned char* ws = nullptr) {
    if (row_parts < 1 || row_part < 0 || row_part >= row_parts) return fail(RT_ERR_INVALID, "bad interleave %d/%d", row_part, row_parts);
    if (!sc || !cam) return fail(RT_ERR_INVALID, "NULL scene or camera");
    if (w <= 0 || h <= 0 || x0 < 0 || y0 < 0 || x0 + w > cam->image_width || y0 + h > cam->image_height)
        return fail(RT_ERR_INVALID, "tile [%d,%d %dx%d] outside the %dx%d image", x0, y0, w, h, cam->image_width, cam->image_height);
    if (!d_rgb || stride < (size_t)w * 3) return fail(RT_ERR_INVALID, "bad output buffer / stride");
    const int N = cam->num_samples, sps = cam->samples_per_side;
    if (N < 1 || sps < 1 || N > 65535) return fail(RT_ERR_INVALID, "num_samples %d / samples_per_side %d not supported", N, sps);
    int depth = sc->ds.max_depth;
    if (depth >= RT_MAX_GEN) return fail(RT_ERR_UNSUPPORTED, "MaxRecursionDepth %d exceeds %d", depth, RT_MAX_GEN - 1);
    HIPCHK(hipSetDevice(sc->device));
    { int arc = check_async(sc); if (arc) return arc; }
    if (depth < 0) {   // ComputeColor returns (0,0,0) for every primary ray (raytracer.cpp:384-386)
        HIPCHK(hipMemset2DAsync(d_rgb, stride, 0, (size_t)w * 3, h, st));
        if (d_hits) HIPCHK(hipMemsetAsync(d_hits, 0xFF, (size_t)w * h * 8, st));
        if (stats) { std::memset(stats, 0, sizeof(*stats)); HIPCHK(hipStreamSynchronize(st)); }
        return RT_OK;
    }
    const int D = sc->has_reflective ? depth : 0;   // without reflective materials no child ray is ever spawned
    int rc;
    const int blocks_closest = ws ? std::min(sc->blocks_closest, RT_NOALLOC_BLOCKS) : sc->blocks_closest;
    const int blocks_shadow = ws ? std::min(sc->blocks_shadow, RT_NOALLOC_BLOCKS) : sc->blocks_shadow;
    // per-wave scratch of the trace loops, sized for the largest resident grid of any kernel that runs them
    const size_t max_waves = ws ? (size_t)std::max(blocks_closest, blocks_shadow) * (RT_TRACE_BLOCK / 64)
                                : std::max<size_t>((size_t)std::max(sc->blocks_closest, sc->blocks_shadow) * (RT_TRACE_BLOCK / 64),
                                                   (size_t)sc->blocks_ordered * (RT_ORDERED_BLOCK / 64));
    const size_t spill_bytes = max_waves * (sc->stack_cap - RT_STACK_LDS) * 64 * sizeof(uint2);
    // candidate hits of rays whose subtrees were traced by other lanes (rt_traverse.h): 64 x RT_CAND_MAX per wave
    const size_t cand_bytes = max_waves * 64 * RT_CAND_MAX * sizeof(uint4);
    Control* ctl;
    uint2* spill;
    uint4* cand;
    Pool ws_pool{};
    uint32_t* ws_mt = nullptr;
    size_t ws_mt_bytes = 0;
    if (ws) {
        if (sc->needs_sequential_rng)
            return fail(RT_ERR_UNSUPPORTED, "rt_render_tile_device_noalloc: scenes whose shading draws random numbers (area lights, rough materials) need per-thread frame stacks; use rt_render_tile_device");
        if ((uint64_t)w * h * N > RT_NOALLOC_MAX_SAMPLES)
            return fail(RT_ERR_INVALID, "rt_render_tile_device_noalloc: %llu pixel samples exceed the maximum of %u", (unsigned long long)w * h * N, RT_NOALLOC_MAX_SAMPLES);
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        size_t off = 0;
        ctl = (Control*)(ws + off); off += up(sizeof(Control));
        spill = (uint2*)(ws + off); off += up(spill_bytes);
        cand = (uint4*)(ws + off); off += up(cand_bytes);
        if (off + (1u << 20) > RT_NOALLOC_SCRATCH_BYTES)
            return fail(RT_ERR_UNSUPPORTED, "rt_render_tile_device_noalloc: the scene's traversal stacks (%u entries per ray) do not fit the %u MiB static workspace", sc->stack_cap, RT_NOALLOC_SCRATCH_BYTES >> 20);
        // the rest: records (104 B) + shadow slots (5 B each) of one chunk, and for N > 1 the generator state of its pixels
        const size_t avail = RT_NOALLOC_SCRATCH_BYTES - off - 4096;
        const size_t lights = std::min<size_t>(std::max<uint32_t>(sc->n_point_lights, 1), 4);
        const size_t rec_bytes = 104 + 5 * lights;
        const size_t per_pixel = (size_t)N * (D > 0 ? 2 : 1) * rec_bytes + (N > 1 ? (624 + (size_t)N) * 4 : 0);
        const size_t pixels = (avail / per_pixel) & ~(size_t)63;
        if (pixels < 64) return fail(RT_ERR_INVALID, "rt_render_tile_device_noalloc: one 8x8 tile at %d spp does not fit the static workspace", N);
        const size_t R = pixels * N * (D > 0 ? 2 : 1), S = R * lights;
        Pool& P = ws_pool;
        P.ray0 = (float4*)(ws + off); off += R * 16;
        P.ray1 = (float4*)(ws + off); off += R * 16;
        P.hit0 = (uint4*)(ws + off); off += R * 16;
        P.nrm = (float4*)(ws + off); off += R * 16;
        P.link = (uint4*)(ws + off); off += R * 16;
        P.color = (float4*)(ws + off); off += R * 16;
        P.hit1 = (float2*)(ws + off); off += R * 8;
        P.shadow_rec = (uint32_t*)(ws + off); off += S * 4;
        P.shadow_lit = (uint8_t*)(ws + off); off += up(S);
        P.cap_records = (uint32_t)R;
        P.cap_shadow = (uint32_t)S;
        if (N > 1) { ws_mt = (uint32_t*)(ws + off); ws_mt_bytes = pixels * (624 + (size_t)N) * 4; off += ws_mt_bytes; }
        if (off > RT_NOALLOC_SCRATCH_BYTES) return fail(RT_ERR_HIP, "internal: static workspace layout overruns (%zu bytes)", off);
    } else {
        if ((rc = sc->b_ctl.ensure(sizeof(Control)))) return rc;
        ctl = (Control*)sc->b_ctl.p;
        if ((rc = sc->b_spill.ensure(spill_bytes))) return rc;
        spill = (uint2*)sc->b_spill.p;
        if ((rc = sc->b_cand.ensure(cand_bytes))) return rc;
        cand = (uint4*)sc->b_cand.p;
    }

    const bool want_sync = stats != nullptr || (!ws && (sc->opt_flags & RT_FLAG_TIMING)) || D > 0 || sc->needs_sequential_rng;
    Timer tm{sc, st, !ws && (sc->opt_flags & RT_FLAG_TIMING) != 0};
    const bool near = (sc->opt_flags & RT_FLAG_NEAR_FIRST) != 0 || sc->rebuilt;   // non-parity visiting order (rt.h)
    HIPCHK(hipMemsetAsync(ctl, 0, sizeof(Control), st));

    const hipEvent_t ev_total_a = sc->ev_total_a, ev_total_b = sc->ev_total_b;
    if (stats) HIPCHK(hipEventRecord(ev_total_a, st));

    DCamera dc;
    to_dcamera(cam, dc);
    const float inv_n = 1.0f / N;   // raytracer.cpp:38
    const int tiles_per_row = (w + 7) / 8, tile_rows_all = (h + 7) / 8;
    const int tile_rows = row_part < tile_rows_all ? (tile_rows_all - row_part + row_parts - 1) / row_parts : 0;
    const uint32_t total_tiles = (uint32_t)tiles_per_row * tile_rows;
    // Which pipeline (rt_mega.h): all generations of a chunk in ONE launch for renders of up to RT_MEGA_MAX_PRIMARIES
    // primary rays — launch count and end games dominate those — and the per-bounce pipeline for larger ones, whose
    // throughput is higher (DESIGN.md §5 has the measurements).  RT_FLAG_PER_BOUNCE_LAUNCHES / RT_FLAG_ONE_LAUNCH
    // force either.  The no-allocation variant keeps the per-bounce pipeline: its fixed pool has no room for the
    // shading waves' slabs, and the rings are 400 MB.
    uint64_t mega_max = RT_MEGA_MAX_PRIMARIES;
    if (const char* e = getenv("RT_MEGA_MAX_PRIMARIES")) mega_max = strtoull(e, nullptr, 10);
    bool mega = !ws && !sc->needs_sequential_rng && !(sc->opt_flags & (RT_FLAG_PER_BOUNCE_LAUNCHES | RT_FLAG_SEPARATE_LAUNCHES)) &&
                      (sc->n_point_lights != 0 || D > 0) &&
                      ((sc->opt_flags & RT_FLAG_ONE_LAUNCH) || (uint64_t)total_tiles * 64 * N <= mega_max);
    if (mega) {
        if ((rc = sc->b_wq.ensure((size_t)blocks_closest * (RT_MEGA_CQ_CAP + RT_MEGA_AQ_CAP) * 4))) return rc;
        if ((rc = sc->b_sq.ensure((size_t)blocks_closest * RT_MEGA_TRACERS * RT_MEGA_SQ_CAP * 4))) return rc;
    }
    // what the shading waves' slabs may hold without using it: one slab per workgroup, and up to a batch's worth (64
    // records with two children each; 64 x lights slots) at the end of every slab, a sixteenth of it
    const uint64_t mega_slack_rec = (uint64_t)blocks_closest * RT_MEGA_SHADERS * RT_MEGA_REC_SLAB;
    const uint64_t mega_slack_sh = (uint64_t)blocks_closest * RT_MEGA_SHADERS * std::max<uint64_t>(RT_MEGA_SH_SLAB, 1024ull * sc->n_point_lights);
    if (!sc->needs_sequential_rng && !ws) {
        // records: one per primary sample, and with child rays about twice that (a chunk that overflows is redone
        // with half the pixels): room for four per sample of the largest chunk
        const uint64_t chunk = std::min<uint64_t>((uint64_t)total_tiles * 64 * N, sc->opt_chunk_samples);
        if ((rc = ensure_pool(sc, D > 0 ? 4 * chunk : chunk, mega ? mega_slack_rec : 0, mega ? mega_slack_sh : 0))) return rc;
    }
    const Pool& P = ws ? ws_pool : sc->pool;
    // (a pool capped — RT_OPT_POOL_RECORDS — below the shading waves' slabs plus the smallest pool ensure_pool makes: per bounce)
    if (mega && (uint64_t)P.cap_records < mega_slack_rec + (1ull << 16)) mega = false;
    // tiles per chunk: bounded by the chunk option, the record pool and the shadow pool
    const uint64_t slack_rec = mega ? mega_slack_rec : 0, slack_sh = mega ? mega_slack_sh : 0;
    const uint64_t usable_rec = P.cap_records > slack_rec ? P.cap_records - slack_rec : 0;
    const uint64_t usable_sh = P.cap_shadow > slack_sh ? (mega ? (P.cap_shadow - slack_sh) / 17 * 16 : P.cap_shadow) : 0;
    uint64_t max_samples = std::min<uint64_t>(sc->opt_chunk_samples, D > 0 ? usable_rec / 2 : usable_rec);
    if (sc->n_point_lights) max_samples = std::min<uint64_t>(max_samples, usable_sh / sc->n_point_lights);
    uint32_t tiles_per_chunk = (uint32_t)std::max<uint64_t>(1, max_samples / (64ull * N));
    if (ws && N > 1) tiles_per_chunk = std::min<uint32_t>(tiles_per_chunk, (uint32_t)(ws_mt_bytes / (64 * (624 + (size_t)N) * 4)));
    if (!sc->needs_sequential_rng && (uint64_t)tiles_per_chunk * 64 * N > P.cap_records) return fail(RT_ERR_OVERFLOW, "record pool too small for one 8x8 tile at %d spp", N);

    uint32_t chunks = 0, retries = 0;
    uint64_t peak = 0;
    unsigned long long good_rays[3] = {0, 0, 0};     // ray counters after the last chunk that was kept
    uint32_t tile = 0;
    if (sc->needs_sequential_rng) {
        // Shading draws random numbers (glossy materials, area lights): one lane per pixel walks the recursion
        // depth-first so that the draws come in the reference's order (rt_ordered.h).  One launch, grid-stride.
        const uint32_t n_idx = total_tiles * 64u;
        const int grid = (int)std::min<uint32_t>((uint32_t)sc->blocks_ordered, (n_idx + RT_ORDERED_BLOCK - 1) / RT_ORDERED_BLOCK);
        // (a lane owns one pixel at a time — generator, frame stack, loop state — and RT_ORDERED_REQS request slots)
        const size_t threads = (size_t)std::max(grid, 1) * RT_ORDERED_BLOCK;
        if ((rc = sc->b_mt.ensure(threads * (624 + (size_t)N) * 4))) return rc;
        if ((rc = sc->b_frames.ensure(threads * (size_t)(depth + 1) * OF_COUNT * 4))) return rc;
        if ((rc = sc->b_pstate.ensure(threads * RT_ORDERED_REQS * (size_t)PS_COUNT * 4))) return rc;
        OrderedArgs A;
        A.cam = dc;
        A.g = ChunkGeom{x0, y0, w, h, tiles_per_row, 0u, n_idx, row_parts, row_part};
        A.area_lights = (const DAreaLight*)sc->b_area_lights.p;
        A.n_area_lights = sc->n_area_lights;
        A.mt_scratch = (uint32_t*)sc->b_mt.p;
        A.frames = (float*)sc->b_frames.p;
        A.pstate = (uint32_t*)sc->b_pstate.p;
        A.inv_n = inv_n;
        A.rgb = d_rgb;
        A.row_stride = stride;
        A.hit_ids = d_hits;
        A.tri_shade = sc->ds.tri_shade;
        A.tile_order = nullptr;
        // Costliest tiles first (k_tile_cost): one centre ray per pixel through the generation pipeline's first two
        // kernels, the tiles sorted on the host.  Worth it when the launch hands out many more tiles than it has waves.
        bool cost_order = grid > 0 && !getenv("RT_ORDERED_IMAGE_ORDER") &&
                          ((N > 1 && total_tiles >= 4u * (uint32_t)grid * (RT_ORDERED_BLOCK / 64)) || getenv("RT_ORDERED_COST_ORDER"));
        if (cost_order) {
            if ((rc = ensure_pool(sc, n_idx))) return rc;
            // (a record pool capped below one record per pixel — RT_OPT_POOL_RECORDS — keeps image order)
            cost_order = sc->pool.cap_records >= n_idx;
        }
        if (cost_order) {
            if ((rc = sc->b_tile_order.ensure((size_t)total_tiles * 8))) return rc;
            uint32_t* d_cost = (uint32_t*)sc->b_tile_order.p + total_tiles;
            DCamera dc1 = dc;
            dc1.num_samples = 1;
            dc1.sps = 1;
            if ((rc = tm.begin(TC_RAYGEN))) return rc;
            k_raygen_single<<<(n_idx + 255) / 256, 256, 0, st>>>(dc1, A.g, sc->pool, ctl, n_idx);
            if (near) k_trace_closest<true><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, sc->pool, ctl, 0, spill, cand);
            else k_trace_closest<false><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, sc->pool, ctl, 0, spill, cand);
            k_tile_cost<<<(n_idx + 255) / 256, 256, 0, st>>>(sc->ds, sc->pool, n_idx, d_cost);
            std::vector<uint32_t> cost(total_tiles), order(total_tiles);
            HIPCHK(hipMemcpyAsync(cost.data(), d_cost, (size_t)total_tiles * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (uint32_t t = 0; t < total_tiles; t++) order[t] = t;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
            HIPCHK(hipMemcpyAsync(sc->b_tile_order.p, order.data(), (size_t)total_tiles * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));                  // (`order` leaves scope)
            HIPCHK(hipMemsetAsync(ctl, 0, sizeof(Control), st));   // cursors and ray counters start again
            if ((rc = tm.end())) return rc;
            A.tile_order = (const uint32_t*)sc->b_tile_order.p;
        }
        if (grid > 0) {
            if ((rc = tm.begin(TC_CLOSEST))) return rc;
            if (near) k_pixel_ordered<true><<<grid, RT_ORDERED_BLOCK, 0, st>>>(sc->ds, A, ctl, spill, cand);
            else k_pixel_ordered<false><<<grid, RT_ORDERED_BLOCK, 0, st>>>(sc->ds, A, ctl, spill, cand);
            if ((rc = tm.end())) return rc;
            HIPCHK(hipGetLastError());
        }
        chunks = 1;
        peak = n_idx;
        tile = total_tiles;
    }
    while (tile < total_tiles) {
        const uint32_t nt = std::min(tiles_per_chunk, total_tiles - tile);
        ChunkGeom g{x0, y0, w, h, tiles_per_row, tile, nt * 64u, row_parts, row_part};
        const uint32_t n_primary = g.n_idx * (uint32_t)N;
        if ((rc = tm.begin(TC_RAYGEN))) return rc;
        if (N == 1) {
            k_raygen_single<<<(g.n_idx + 255) / 256, 256, 0, st>>>(dc, g, P, ctl, n_primary);
        } else {
            const size_t strideT = g.n_idx;
            if (!ws && (rc = sc->b_mt.ensure(strideT * (624 + (size_t)N) * 4))) return rc;
            uint32_t* mt = ws ? ws_mt : (uint32_t*)sc->b_mt.p;
            k_raygen_multi<<<(g.n_idx + 255) / 256, 256, 0, st>>>(dc, g, P, mt, mt + strideT * 624, strideT, ctl, n_primary);
        }
        if ((rc = tm.end())) return rc;
        const int grid_flat = 2048;
        // closest(0); then per generation: shade(d), and ONE launch for what shade(d) produced — the shadow slots
        // of generation d and the child rays of generation d + 1 are independent of each other (k_trace_mixed)
        const bool merge = !(sc->opt_flags & RT_FLAG_SEPARATE_LAUNCHES);
        // (generation 0 of a one-sample frame is combined in k_resolve, by the thread that writes the pixel; with several samples
        // per pixel that thread would combine them one after the other — 16 spp: 8.1 -> 13.8 ms for the two stages)
        const bool combine0_in_resolve = !mega && cam->num_samples == 1;
        if (mega) {
            // one launch traces every ray of the chunk: primaries, the children its shading waves spawn, the shadow slots
            // tracing waves that take shadow slots first: their share of the work, about 0.7 L h / (1 + 0.7 L h) for L lights and
            // a hit fraction h ~ 0.7 (an any-hit ray costs ~0.7 of a closest-hit ray); RT_MEGA_ANY_WAVES overrides (experiments)
            const double sh = 0.5 * (double)sc->n_point_lights;
            uint32_t any_waves = sc->n_point_lights ? (uint32_t)std::lround(RT_MEGA_TRACERS * sh / (1.0 + sh)) : 0u;
            if (const char* e = getenv("RT_MEGA_ANY_WAVES")) any_waves = (uint32_t)atoi(e);
            any_waves = std::min<uint32_t>(any_waves, RT_MEGA_TRACERS - 1);
            const MegaArgs ma{(uint32_t*)sc->b_wq.p, (uint32_t*)sc->b_wq.p + (size_t)blocks_closest * RT_MEGA_CQ_CAP, (uint32_t*)sc->b_sq.p, n_primary, any_waves,
                              getenv("RT_MEGA_STRICT") ? (uint32_t)atoi(getenv("RT_MEGA_STRICT")) : 0u};
            if ((rc = tm.begin(TC_CLOSEST))) return rc;
            if (near) k_trace_mega<true><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, ma, spill, cand);
            else k_trace_mega<false><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, ma, spill, cand);
            if ((rc = tm.end())) return rc;
            if ((rc = tm.begin(TC_COMBINE))) return rc;
            for (int d = D; d >= 0; d--) k_combine_mega<<<grid_flat, 256, 0, st>>>(sc->ds, P, ctl, d, n_primary);
            if ((rc = tm.end())) return rc;
        } else {
        if ((rc = tm.begin(TC_CLOSEST))) return rc;
        if (near) k_trace_closest<true><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, 0, spill, cand);
        else k_trace_closest<false><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, 0, spill, cand);
        if ((rc = tm.end())) return rc;
        for (int d = 0; d <= D; d++) {
            if ((rc = tm.begin(TC_SHADE))) return rc;
            k_shade<<<grid_flat, 256, 0, st>>>(sc->ds, P, ctl, d);
            k_advance<<<1, 64, 0, st>>>(ctl, P, d);
            if ((rc = tm.end())) return rc;
            if (d < D && sc->n_point_lights && merge) {
                if ((rc = tm.begin(TC_CLOSEST))) return rc;      // (timed under "closest": both kinds of ray)
                if (near) k_trace_mixed<true><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, d, d + 1, spill, cand);
                else k_trace_mixed<false><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, d, d + 1, spill, cand);
                if ((rc = tm.end())) return rc;
                continue;
            }
            if (sc->n_point_lights) {
                if ((rc = tm.begin(TC_SHADOW))) return rc;
                k_trace_shadow<<<blocks_shadow, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, d, spill, cand);
                if ((rc = tm.end())) return rc;
            }
            if (d < D) {
                if ((rc = tm.begin(TC_CLOSEST))) return rc;
                if (near) k_trace_closest<true><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, d + 1, spill, cand);
                else k_trace_closest<false><<<blocks_closest, RT_TRACE_BLOCK, 0, st>>>(sc->ds, P, ctl, d + 1, spill, cand);
                if ((rc = tm.end())) return rc;
            }
        }
        if ((rc = tm.begin(TC_COMBINE))) return rc;
        for (int d = D; d >= (combine0_in_resolve ? 1 : 0); d--) k_combine<<<grid_flat, 256, 0, st>>>(sc->ds, P, ctl, d);
        if ((rc = tm.end())) return rc;
        }
        if ((rc = tm.begin(TC_RESOLVE))) return rc;
        if (!combine0_in_resolve) k_resolve<false><<<(g.n_idx + 255) / 256, 256, 0, st>>>(sc->ds, dc, g, P, sc->ds.tri_shade, inv_n, d_rgb, stride, d_hits);
        else k_resolve<true><<<(g.n_idx + 255) / 256, 256, 0, st>>>(sc->ds, dc, g, P, sc->ds.tri_shade, inv_n, d_rgb, stride, d_hits);
        if ((rc = tm.end())) return rc;
        HIPCHK(hipGetLastError());
        chunks++;

        if (D > 0) {
            // child rays were spawned: the pools may have overflowed
            HIPCHK(hipMemcpyAsync(sc->h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (sc->h_ctl->overflow & ~3u) return ctl_status(sc->h_ctl->overflow & ~3u, sc->stack_cap);
            if (sc->h_ctl->overflow & 3u) {
                if (tiles_per_chunk == 1) return fail(RT_ERR_OVERFLOW, "ray pool too small for one 8x8 tile; raise RT_OPT_POOL_RECORDS");
                tiles_per_chunk = std::max(1u, tiles_per_chunk / 2);
                // redo this chunk with fewer pixels; the rays of the failed try do not count
                HIPCHK(hipMemcpyAsync(&ctl->rays_primary, good_rays, sizeof(good_rays), hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                retries++;
                continue;
            }
            good_rays[0] = sc->h_ctl->rays_primary; good_rays[1] = sc->h_ctl->rays_secondary; good_rays[2] = sc->h_ctl->rays_shadow;
            peak = std::max<uint64_t>(peak, sc->h_ctl->n_records);
        } else {
            peak = std::max<uint64_t>(peak, n_primary);
        }
        tile += nt;
    }

    if (stats) HIPCHK(hipEventRecord(ev_total_b, st));
    if (want_sync) {
        HIPCHK(hipMemcpyAsync(sc->h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if ((rc = ctl_status(sc->h_ctl->overflow, sc->stack_cap))) return rc;
    } else {
        // asynchronous return: what the kernels report (stack / shadow-pool overflow, scheduler bounds) is copied
        // behind them and looked at by rt_sync or the next call on this scene (check_async) — never dropped
        HIPCHK(hipMemcpyAsync(sc->h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (st) (void)hipStreamIsCapturing(st, &cap);
        if (cap == hipStreamCaptureStatusActive) {
            // recorded into a graph (rt.h, rt_render_tile_device_noalloc): an event recorded here cannot be waited for
            // outside the graph; every replay copies its Control block to h_ctl, and rt_sync reads that
            sc->captured_check = true;
            sc->h_ctl->overflow = 0;
        } else {
            HIPCHK(hipEventRecord(sc->ev_async, st));
            sc->pending_check = true;
        }
    }
#ifdef RT_TIMELINE
    if (want_sync) {
        static uint32_t tlw[RT_TL_WAVES][RT_TL_BINS][8];
        static unsigned long long tl[RT_TL_BINS][8];
        if (hipMemcpyFromSymbol(tlw, HIP_SYMBOL(g_tl), sizeof(tlw)) == hipSuccess) {
            std::memset(tl, 0, sizeof(tl));
            for (int w = 0; w < RT_TL_WAVES; w++) for (int b = 0; b < RT_TL_BINS; b++) for (int k = 0; k < 8; k++) tl[b][k] += tlw[w][b][k];
            for (int b = 0; b < RT_TL_BINS; b++)
                if (tl[b][0])
                    fprintf(stderr, "[rt tl] %5d us: rounds %6llu slots/round %5.1f helpers/round %4.1f | node steps %7llu lanes/step %5.1f | tri steps %6llu lanes/step %5.1f | drained %.2f\n", b * 10, tl[b][0],
                            (double)tl[b][1] / tl[b][0], (double)tl[b][6] / tl[b][0], tl[b][2], tl[b][2] ? (double)tl[b][3] / tl[b][2] : 0.0, tl[b][4], tl[b][4] ? (double)tl[b][5] / tl[b][4] : 0.0, (double)tl[b][7] / tl[b][0]);
            std::memset(tlw, 0, sizeof(tlw));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), tlw, sizeof(tlw));
            unsigned long long tt[8] = {0};
            if (hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_tl_tail), sizeof(tt)) == hipSuccess && tt[5]) {
                fprintf(stderr, "[rt tl tail] %llu rounds of waves with rays after the drain: per round sched %.2f node %.2f tri %.2f big %.2f trans+refill %.2f us\n", tt[5],
                        tt[0] * 0.01 / tt[5], tt[1] * 0.01 / tt[5], tt[2] * 0.01 / tt[5], tt[3] * 0.01 / tt[5], tt[4] * 0.01 / tt[5]);
                unsigned long long t2[8] = {0};
                if (hipMemcpyFromSymbol(t2, HIP_SYMBOL(g_tl_tail2), sizeof(t2)) == hipSuccess) {
                    fprintf(stderr, "[rt tl tail] per round: sched = claim %.2f + announce %.2f + post %.2f + rest %.2f us; transitions = block %.2f + retire %.2f us (all passes) | %llu replays, %.2f candidates each\n",
                            t2[0] * 0.01 / tt[5], t2[1] * 0.01 / tt[5], t2[2] * 0.01 / tt[5], t2[3] * 0.01 / tt[5], t2[4] * 0.01 / tt[5], t2[5] * 0.01 / tt[5],
                            t2[6], t2[6] ? (double)t2[7] / t2[6] : 0.0);
                    std::memset(t2, 0, sizeof(t2));
                    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_tail2), t2, sizeof(t2));
                }
                std::memset(tt, 0, sizeof(tt));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_tail), tt, sizeof(tt));
            }
            static uint32_t lg[RT_TL_LOG][8];
            uint32_t ln = 0;
            if (hipMemcpyFromSymbol(&ln, HIP_SYMBOL(g_tl_log_n), 4) == hipSuccess && hipMemcpyFromSymbol(lg, HIP_SYMBOL(g_tl_log), sizeof(lg)) == hipSuccess) {
                ln = std::min<uint32_t>(ln, RT_TL_LOG);
                std::vector<uint32_t> order(ln);
                for (uint32_t i = 0; i < ln; i++) order[i] = i;
                const int key = getenv("RT_TL_BY_LIFE") ? 2 : 6;   // the longest-lived instead of the last to finish
                std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lg[a][key] > lg[b][key]; });
                fprintf(stderr, "[rt tl] %u rays finished after the cursor was drained; the last 40 to finish:\n", ln);
                for (uint32_t i = 0; i < ln && i < 40; i++) {
                    const uint32_t* o = lg[order[i]];
                    fprintf(stderr, "[rt tl ray] end@%7.1f us life %7.1f us | own node steps %5u leaves %4u | posts %3u wait-rounds %4u rounds %4u | %.2f us/step | item %u\n", o[6] * 0.01, o[2] * 0.01, o[0], o[1], o[3], o[4], o[5],
                            o[0] ? o[2] * 0.01 / o[0] : 0.0, o[7]);
                }
                // by lifetime
                unsigned long long hist[16] = {0}, steps[16] = {0};
                for (uint32_t i = 0; i < ln; i++) { int b = 0; uint32_t us = lg[i][2] / 100; while (us > 1 && b < 15) { us >>= 1; b++; } hist[b]++; steps[b] += lg[i][0]; }
                for (int b = 0; b < 16; b++) if (hist[b]) fprintf(stderr, "[rt tl life] %6d us+: %7llu rays, mean own node steps %.1f\n", 1 << b, hist[b], (double)steps[b] / hist[b]);
            }
            ln = 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_log_n), &ln, 4);
            const unsigned long long big = ~0ull;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_t0), &big, sizeof(big));
        }
    }
#endif
#ifdef RT_TRACE_STATS
    if (want_sync) {
        const unsigned long long* d = sc->h_ctl->dbg;
        fprintf(stderr, "[rt stats] closest: wave_node_steps=%llu lane_node_steps=%llu (util %.3f) wave_tri_steps=%llu lane_tri_steps=%llu (util %.3f) "
                        "outer=%llu rays=%llu | per ray: node %.1f tri %.2f | per 64-ray batch: wave node steps %.1f tri steps %.1f outer %.1f\n",
                d[0], d[1], d[0] ? d[1] / (64.0 * d[0]) : 0.0, d[2], d[3], d[2] ? d[3] / (64.0 * d[2]) : 0.0, d[4], d[5],
                d[5] ? (double)d[1] / d[5] : 0.0, d[5] ? (double)d[3] / d[5] : 0.0, d[5] ? 64.0 * d[0] / d[5] : 0.0,
                d[5] ? 64.0 * d[2] / d[5] : 0.0, d[5] ? 64.0 * d[4] / d[5] : 0.0);
        // wall_clock64 ticks are 10 ns
        {
            static unsigned long long hw[8192][16];
            if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wave_dbg), sizeof(hw)) == hipSuccess) {
                const int nw = std::min(8192, sc->blocks_closest * (RT_TRACE_BLOCK / 64));
                std::vector<int> order(nw);
                for (int i = 0; i < nw; i++) order[i] = i;
                std::sort(order.begin(), order.end(), [&](int a, int b) { return hw[a][0] > hw[b][0]; });
                double sum[16] = {0};
                for (int i = 0; i < nw; i++) for (int k = 0; k < 15; k++) sum[k] += (double)hw[i][k];
                fprintf(stderr, "[rt stats] per-wave mean: life %.0f us exhausted@%.0f | sched %.0f node %.0f tri %.0f big %.0f trans %.0f us | outer %.0f nodesteps %.0f tristeps %.0f donations %.1f coop %.1f rays %.0f\n",
                        sum[0] * 0.01 / nw, sum[1] * 0.01 / nw, sum[2] * 0.01 / nw, sum[3] * 0.01 / nw, sum[4] * 0.01 / nw, sum[5] * 0.01 / nw, sum[6] * 0.01 / nw,
                        sum[7] / nw, sum[8] / nw, sum[9] / nw, sum[10] / nw, sum[11] / nw, sum[12] / nw);
                for (int r = 0; r < 6 && r < nw; r++) {
                    const unsigned long long* o = hw[order[r]];
                    fprintf(stderr, "[rt stats] wave %d: life %.0f us exhausted@%.0f | sched %.0f node %.0f tri %.0f big %.0f trans %.0f us | outer %llu nodesteps %llu (lanes %llu) tristeps %llu (lanes %llu) donations %llu coop %llu rays %llu\n",
                            order[r], o[0] * 0.01, o[1] * 0.01, o[2] * 0.01, o[3] * 0.01, o[4] * 0.01, o[5] * 0.01, o[6] * 0.01, o[7], o[8], o[13], o[9], o[14], o[10], o[11], o[12]);
                }
            }
        }
        {   // chip-wide activity over time: wave steps and active lanes per 25-us bin (both trace kernels of the frame add up)
            static unsigned long long hh[RT_HIST_BINS][4];
            if (hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_hist), sizeof(hh)) == hipSuccess) {
                for (int b = 0; b < RT_HIST_BINS; b++)
                    if (hh[b][0] | hh[b][2])
                        fprintf(stderr, "[rt hist] %5d us: node steps %7llu util %.2f | tri steps %7llu util %.2f\n", b * 25, hh[b][0],
                                hh[b][0] ? hh[b][1] / (64.0 * hh[b][0]) : 0.0, hh[b][2], hh[b][2] ? hh[b][3] / (64.0 * hh[b][2]) : 0.0);
                std::memset(hh, 0, sizeof(hh));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hist), hh, sizeof(hh));
            }
        }
        {
            static unsigned long long rh[2][20];
            if (hipMemcpyFromSymbol(rh, HIP_SYMBOL(g_rayhist), sizeof(rh)) == hipSuccess) {
                for (int k = 0; k < 2; k++) {
                    fprintf(stderr, "[rt rayhist] %s per ray (0, 1, 2-3, 4-7, ...):", k ? "node steps" : "leaves");
                    for (int b = 0; b < 20; b++) fprintf(stderr, " %llu", rh[k][b]);
                    fprintf(stderr, "\n");
                }
                std::memset(rh, 0, sizeof(rh));
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rayhist), rh, sizeof(rh));
            }
        }
        fprintf(stderr, "[rt stats] transitions, summed over waves: %.0f us = block %.0f + retire %.0f + refill %.0f us in %llu passes\n",
                d[10] * 0.01, d[6] * 0.01, d[7] * 0.01, d[8] * 0.01, d[9]);
        {
            unsigned long long y[8] = {0}, z[8] = {0};
            if (hipMemcpyFromSymbol(y, HIP_SYMBOL(g_ydbg), sizeof(y)) == hipSuccess) {
                fprintf(stderr, "[rt stats] transition block: mesh end %.0f us, prims loop %.0f us; bodies run %llu (lanes %llu), prim iterations %llu (lanes %llu)\n",
                        y[0] * 0.01, y[1] * 0.01, y[4], y[5], y[2], y[3]);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ydbg), z, sizeof(z));
            }
        }
    }
#endif
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_primary = sc->h_ctl->rays_primary;
        stats->rays_secondary = sc->h_ctl->rays_secondary;
        stats->rays_shadow = sc->h_ctl->rays_shadow;
        stats->records_peak = peak;
        stats->chunks = chunks - retries;     // chunks kept (a chunk that overflowed its pool is redone smaller)
        HIPCHK(hipEventElapsedTime(&stats->ms_total, ev_total_a, ev_total_b));
    }
    // fold kernel timings
    float acc[TC_COUNT] = {0};
    uint32_t launches[TC_COUNT] = {0};
    for (auto& e : sc->ev_used) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { acc[e.cat] += ms; launches[e.cat]++; }
        sc->ev_free.push_back(e);
    }
    sc->ev_used.clear();
    if (stats) {
        stats->ms_raygen = acc[TC_RAYGEN]; stats->ms_closest = acc[TC_CLOSEST]; stats->ms_shade = acc[TC_SHADE];
        stats->ms_shadow = acc[TC_SHADOW]; stats->ms_combine = acc[TC_COMBINE]; stats->ms_resolve = acc[TC_RESOLVE];
        stats->launches_closest = launches[TC_CLOSEST];
        stats->launches_shadow = launches[TC_SHADOW];
    }
    return RT_OK;
}

}  // namespace

// rt.h: nothing throws across the boundary.  The host side allocates (std::vector, std::thread); a failure there
// becomes a status like any other.
template <class F>
static int guarded(F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID, "internal error: %s", e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID, "internal error");
    }
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int rt_upload_scene(int device, const rt_scene_desc* scene, rt_scene** out) { return rt_upload_scene_ex(device, scene, 0u, out); }

int rt_upload_scene_ex(int device, const rt_scene_desc* scene, uint32_t upload_flags, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    if (upload_flags & ~(uint32_t)(RT_UPLOAD_REBUILD_MESH_BVH | RT_UPLOAD_REBUILD_ON_DEVICE | RT_UPLOAD_REBUILD_LBVH)) return fail(RT_ERR_INVALID, "unknown upload flag");
    if ((upload_flags & RT_UPLOAD_REBUILD_ON_DEVICE) && !(upload_flags & RT_UPLOAD_REBUILD_MESH_BVH)) return fail(RT_ERR_INVALID, "RT_UPLOAD_REBUILD_ON_DEVICE goes with RT_UPLOAD_REBUILD_MESH_BVH");
    if ((upload_flags & RT_UPLOAD_REBUILD_LBVH) && !(upload_flags & RT_UPLOAD_REBUILD_ON_DEVICE)) return fail(RT_ERR_INVALID, "RT_UPLOAD_REBUILD_LBVH goes with RT_UPLOAD_REBUILD_ON_DEVICE");
    *out = nullptr;
    // the description is checked first (host work only), so a malformed scene is reported as such on any box
    PhaseClock pc;
    int rc = guarded([&] { return validate_scene(scene); });
    if (rc) return rc;
    pc.lap("validate_scene");
    int n = rt_device_count();
    if (n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(RT_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    rt_scene* sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_OOM, "host allocation failed");
    sc->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) sc->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking) != hipSuccess) { delete sc; return fail(RT_ERR_HIP, "hipStreamCreate failed"); }
    if (hipHostMalloc((void**)&sc->h_ctl, sizeof(Control)) != hipSuccess) { rt_free_scene(sc); return fail(RT_ERR_HIP, "hipHostMalloc failed"); }
    if (hipEventCreate(&sc->ev_total_a) != hipSuccess || hipEventCreate(&sc->ev_total_b) != hipSuccess ||
        hipEventCreateWithFlags(&sc->ev_async, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sc->ev_gather, hipEventDisableTiming) != hipSuccess) { rt_free_scene(sc); return fail(RT_ERR_HIP, "hipEventCreate failed"); }
    rc = guarded([&] { return build_device_scene(sc, scene, upload_flags); });
    if (rc) { rt_free_scene(sc); return rc; }
    if (device < RT_MAX_DEVICES) {
        // this GPU's copy of the static workspace (rt_render_tile_device_noalloc): asked for once per device, with
        // that device current (hipSetDevice above); every GPU has its own copy at its own address
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            unsigned char* ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) { rt_free_scene(sc); return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e)); }
            g_scratch_ptr[device] = ptr;
        }
        sc->static_ws = g_scratch_ptr[device];
    }
    // persistent grids: as many workgroups as are resident at once (RT_TRACE_BLOCKS_PER_CU caps it, for occupancy experiments)
    int per_cu = 0, cap = 8;
    if (const char* e = getenv("RT_TRACE_BLOCKS_PER_CU")) cap = std::max(1, std::min(8, atoi(e)));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_closest<false>, RT_TRACE_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    sc->blocks_closest = sc->n_cu * std::min(per_cu, cap);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_shadow, RT_TRACE_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    sc->blocks_shadow = sc->n_cu * std::min(per_cu, cap);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pixel_ordered<false>, RT_ORDERED_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    sc->blocks_ordered = sc->n_cu * std::min(per_cu, cap);
    *out = sc;
    return RT_OK;
}

int rt_update_scene(rt_scene* sc, const rt_scene_desc* scene, uint32_t flags) {
    if (!sc) return fail(RT_ERR_INVALID, "scene is NULL");
    int rc = guarded([&] { return validate_scene(scene, false); });
    if (rc) return rc;
    const rt_scene::MeshLevel& M = sc->mesh;
    if (scene->n_mesh_bvhs != M.n_mesh_bvhs || scene->n_mesh_nodes != M.n_mesh_nodes || scene->n_mesh_prims != M.n_mesh_prims ||
        scene->n_faces != M.n_faces || scene->n_vertices != M.n_vertices)
        return fail(RT_ERR_INVALID, "rt_update_scene: the mesh arrays differ in size from the uploaded scene's; upload the scene instead");
    if ((flags & RT_UPDATE_VERIFY_MESHES) && mesh_hash(scene) != M.hash)
        return fail(RT_ERR_INVALID, "rt_update_scene: the mesh arrays differ from the uploaded scene's; upload the scene instead");
    HIPCHK(hipSetDevice(sc->device));
    HIPCHK(hipStreamSynchronize(sc->stream));      // no render of this scene may be in flight (the caller's stream: caller's duty)
    { int arc = check_async(sc); if (arc) return arc; }
    return guarded([&] { return build_top_level(sc, scene); });
}

int rt_scene_set_option(rt_scene* s, int option, uint64_t value) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    switch (option) {
    case RT_OPT_FLAGS: s->opt_flags = value; return RT_OK;
    case RT_OPT_POOL_RECORDS:
        if (value < 4096 || value > (1ull << 31)) return fail(RT_ERR_INVALID, "pool size out of range");
        s->opt_pool_records = value;
        return RT_OK;
    case RT_OPT_CHUNK_SAMPLES:
        if (value < 64) return fail(RT_ERR_INVALID, "chunk size out of range");
        s->opt_chunk_samples = value;
        return RT_OK;
    case RT_OPT_STACK_CAP:
        if (value != 0 && (value <= RT_STACK_LDS || value > RT_STACK_CAP_LIMIT)) return fail(RT_ERR_INVALID, "stack capacity out of range");
        s->stack_cap_override = (uint32_t)value;
        s->stack_cap = value ? (uint32_t)value : s->stack_cap_derived;   // 0: the derived capacity again, at once
        s->ds.stack_cap = s->stack_cap;
        return RT_OK;
    default: return fail(RT_ERR_INVALID, "unknown option %d", option);
    }
}

int rt_render_tile_device(rt_scene* s, const rt_camera_desc* cam, int x0, int y0, int tile_w, int tile_h, void* d_rgb_out,
                          size_t rgb_row_stride_bytes, void* d_hit_id_out, rt_stats* stats, void* hip_stream) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
    return guarded([&] { return render_impl(s, cam, x0, y0, tile_w, tile_h, 1, 0, (uint8_t*)d_rgb_out, rgb_row_stride_bytes, (uint32_t*)d_hit_id_out, stats, st); });
}

int rt_render_tile_rows_device(rt_scene* s, const rt_camera_desc* cam, int x0, int y0, int tile_w, int tile_h, int row_parts,
                               int row_part, void* d_rgb_out, size_t rgb_row_stride_bytes, void* d_hit_id_out, rt_stats* stats,
                               void* hip_stream) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
    return guarded([&] {
        return render_impl(s, cam, x0, y0, tile_w, tile_h, row_parts, row_part, (uint8_t*)d_rgb_out, rgb_row_stride_bytes,
                           (uint32_t*)d_hit_id_out, stats, st);
    });
}

int rt_render_tile_device_noalloc(rt_scene* s, const rt_camera_desc* cam, int x0, int y0, int tile_w, int tile_h, void* d_rgb_out,
                                  size_t rgb_row_stride_bytes, void* d_hit_id_out, void* hip_stream) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    if (!s->static_ws) return fail(RT_ERR_UNSUPPORTED, "rt_render_tile_device_noalloc: no static workspace on device %d", s->device);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
    return guarded([&] {
        return render_impl(s, cam, x0, y0, tile_w, tile_h, 1, 0, (uint8_t*)d_rgb_out, rgb_row_stride_bytes, (uint32_t*)d_hit_id_out, nullptr, st, s->static_ws);
    });
}

int rt_render_tile(rt_scene* s, const rt_camera_desc* cam, int x0, int y0, int tile_w, int tile_h, uint8_t* rgb_out,
                   size_t rgb_row_stride_bytes, uint32_t* hit_id_out, rt_stats* stats) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    if (!rgb_out || tile_w <= 0 || tile_h <= 0 || rgb_row_stride_bytes < (size_t)tile_w * 3) return fail(RT_ERR_INVALID, "bad output buffer");
    HIPCHK(hipSetDevice(s->device));
    const size_t fb_bytes = (size_t)tile_w * tile_h * 3, hit_bytes = (size_t)tile_w * tile_h * 8;
    int rc = s->b_fb.ensure(fb_bytes);
    if (rc) return rc;
    if (hit_id_out && (rc = s->b_hits.ensure(hit_bytes))) return rc;
    rc = guarded([&] {
        return render_impl(s, cam, x0, y0, tile_w, tile_h, 1, 0, (uint8_t*)s->b_fb.p, (size_t)tile_w * 3, hit_id_out ? (uint32_t*)s->b_hits.p : nullptr,
                           stats, s->stream);
    });
    if (rc) return rc;
    // device -> pinned staging (one contiguous copy at PCIe rate) -> the caller's rows; a 2-D copy straight into
    // pageable memory took 7.5 ms for a 1920x1080 frame, this takes well under 1 ms
    const size_t need = fb_bytes + (hit_id_out ? hit_bytes : 0);
    if (s->h_stage_bytes < need) {
        if (s->h_stage) (void)hipHostFree(s->h_stage);
        s->h_stage = nullptr;
        s->h_stage_bytes = 0;
        HIPCHK(hipHostMalloc(&s->h_stage, need));
        s->h_stage_bytes = need;
    }
    uint8_t* stage = (uint8_t*)s->h_stage;
    HIPCHK(hipMemcpyAsync(stage, s->b_fb.p, fb_bytes, hipMemcpyDeviceToHost, s->stream));
    if (hit_id_out) HIPCHK(hipMemcpyAsync(stage + fb_bytes, s->b_hits.p, hit_bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const size_t row = (size_t)tile_w * 3;
    if (rgb_row_stride_bytes == row) std::memcpy(rgb_out, stage, fb_bytes);
    else for (int y = 0; y < tile_h; y++) std::memcpy(rgb_out + (size_t)y * rgb_row_stride_bytes, stage + (size_t)y * row, row);
    if (hit_id_out) std::memcpy(hit_id_out, stage + fb_bytes, hit_bytes);
    return RT_OK;
}

// The 8-row bands ty = g, g + n, ... of a `rows`-row rectangle as one strided 2-D copy of the full bands plus, for the
// owner of a shorter last band, one more copy.  pitch = bytes between rows in both buffers, row_bytes = bytes copied per row.
static hipError_t copy_bands(void* dst, const void* src, size_t pitch, size_t row_bytes, int rows, int n, int g, hipStream_t st) {
    const int bands_all = (rows + 7) / 8;
    if (g >= bands_all) return hipSuccess;
    const int mine = (bands_all - g + n - 1) / n;
    const bool last_is_mine = (bands_all - 1) % n == g;
    const int last_rows = rows - (bands_all - 1) * 8;               // 1..8
    const int full = (last_is_mine && last_rows < 8) ? mine - 1 : mine;
    hipError_t e = hipSuccess;
    if (row_bytes == pitch) {
        // rows are contiguous: a band is one run of 8 rows, the GPU's bands one 2-D copy with a pitch of n bands
        const size_t band = 8 * pitch;
        if (full > 0) e = hipMemcpy2DAsync((char*)dst + (size_t)g * band, (size_t)n * band, (const char*)src + (size_t)g * band, (size_t)n * band, band, (size_t)full, hipMemcpyDeviceToDevice, st);
    } else {
        for (int b = 0; b < full && e == hipSuccess; b++) {
            const size_t off = (size_t)(g + b * n) * 8 * pitch;
            e = hipMemcpy2DAsync((char*)dst + off, pitch, (const char*)src + off, pitch, row_bytes, 8, hipMemcpyDeviceToDevice, st);
        }
    }
    if (e == hipSuccess && full < mine) {
        const size_t off = (size_t)(bands_all - 1) * 8 * pitch;
        e = hipMemcpy2DAsync((char*)dst + off, pitch, (const char*)src + off, pitch, row_bytes, (size_t)last_rows, hipMemcpyDeviceToDevice, st);
    }
    return e;
}

int rt_gather_rows(rt_scene* const* per_gpu, int n, void* const* d_fb, size_t row_stride_bytes, int width, int height,
                   void* const* hip_streams, uint8_t* host_rgb) {
    if (!per_gpu || !d_fb || n < 1) return fail(RT_ERR_INVALID, "rt_gather_rows: NULL arrays or n < 1");
    if (width <= 0 || height <= 0 || row_stride_bytes < (size_t)width * 3) return fail(RT_ERR_INVALID, "rt_gather_rows: bad rectangle / stride");
    for (int g = 0; g < n; g++)
        if (!per_gpu[g] || !d_fb[g]) return fail(RT_ERR_INVALID, "rt_gather_rows: scene or frame %d is NULL", g);
    return guarded([&]() -> int {
        rt_scene* s0 = per_gpu[0];
        hipStream_t st0 = (hip_streams && hip_streams[0]) ? (hipStream_t)hip_streams[0] : s0->stream;
        const size_t row_bytes = (size_t)width * 3;
        for (int g = 1; g < n; g++) {
            rt_scene* sg = per_gpu[g];
            hipStream_t st = (hip_streams && hip_streams[g]) ? (hipStream_t)hip_streams[g] : sg->stream;
            HIPCHK(hipSetDevice(sg->device));
            if (sg->device != s0->device) {
                // direct writes into GPU 0's memory (xGMI); "already enabled" is fine, "not possible" leaves the copy to the runtime's staging
                const hipError_t pe = hipDeviceEnablePeerAccess(s0->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();
            }
            const hipError_t e = copy_bands(d_fb[0], d_fb[g], row_stride_bytes, row_bytes, height, n, g, st);
            if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_gather_rows: copy of GPU %d's rows failed: %s", g, hipGetErrorString(e));
            HIPCHK(hipEventRecord(sg->ev_gather, st));
        }
        HIPCHK(hipSetDevice(s0->device));
        for (int g = 1; g < n; g++) HIPCHK(hipStreamWaitEvent(st0, per_gpu[g]->ev_gather, 0));
        if (host_rgb) {
            HIPCHK(hipMemcpy2DAsync(host_rgb, row_bytes, d_fb[0], row_stride_bytes, row_bytes, (size_t)height, hipMemcpyDeviceToHost, st0));
            HIPCHK(hipStreamSynchronize(st0));
        }
        return RT_OK;
    });
}

int rt_debug_fetch(rt_scene* s, int which, uint64_t first, uint64_t count, void* out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "NULL argument");
    const DevBuf* b[7] = {&s->b_ray0, &s->b_ray1, &s->b_hit0, &s->b_nrm, &s->b_link, &s->b_color, &s->b_shadow_lit};
    if (which < 0 || which > 6) return fail(RT_ERR_INVALID, "bad array id");
    const size_t el = which == 6 ? 1 : 16;
    if (!b[which]->p || (first + count) * el > b[which]->bytes) return fail(RT_ERR_INVALID, "range outside the pool");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(out, (const char*)b[which]->p + first * el, count * el, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_sync(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return check_async(s);      // (also waits for a render that was enqueued on a caller's stream)
}

void rt_free_scene(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (DevBuf* b : {&s->b_big, &s->b_nodes, &s->b_tris, &s->b_tri_shade, &s->b_vnormals, &s->b_objects, &s->b_top_prims, &s->b_top_items, &s->b_planes,
                      &s->b_materials, &s->b_lights, &s->b_ray0, &s->b_ray1, &s->b_hit0, &s->b_hit1, &s->b_nrm, &s->b_link,
                      &s->b_color, &s->b_shadow_rec, &s->b_shadow_lit, &s->b_ctl, &s->b_spill, &s->b_cand, &s->b_mt, &s->b_frames, &s->b_pstate, &s->b_tile_order, &s->b_area_lights, &s->b_fb, &s->b_hits, &s->b_tri_rank, &s->b_wq, &s->b_sq})
        b->release();
    for (auto& e : s->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : s->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : {s->ev_total_a, s->ev_total_b, s->ev_async, s->ev_gather}) if (e) (void)hipEventDestroy(e);
    if (s->h_ctl) (void)hipHostFree(s->h_ctl);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // extern "C"
