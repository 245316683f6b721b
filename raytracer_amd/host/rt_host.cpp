// rt_host.cpp — host-side helper library (see include/rt_host.h).
//
// Everything here runs before rt_upload_scene: blob I/O and a programmatic
// scene builder that performs the same host-side preparation the reference
// does in parser.cpp / raytracer.cpp:98-339 (citations at each step).  It is
// built with -ffp-contract=off so its FP32 results match the reference build
// (x86-64 baseline, no FMA).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "rt_host.h"
#include "rt_scene_store.hpp"

namespace {

thread_local std::string g_err;

constexpr float kFltMax = 3.402823466e+38F;   // MathF.h:6

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 ld(const float* p) { return {p[0], p[1], p[2]}; }
inline void st(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// MathF.h:18-28
inline float q_rsqrt(float x) {
    if (x <= 0.0f) return 0.0f;
    uint32_t i;
    std::memcpy(&i, &x, 4);
    i = 0x5f3759dfu - (i >> 1);
    float y;
    std::memcpy(&y, &i, 4);
    return y * (1.5f - 0.5f * x * y * y);
}
// Vec3f.h:30-39
inline V3 normalize(V3 a) {
    float inv = q_rsqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (inv == 0.0f) return {0.f, 0.f, 0.f};
    return a * inv;
}

struct Box {
    V3 mn{kFltMax, kFltMax, kFltMax}, mx{-kFltMax, -kFltMax, -kFltMax};   // Aabb.h:11-15
    void expand(V3 p) {                                                    // Aabb.h:17-26
        mn.x = std::min(mn.x, p.x); mn.y = std::min(mn.y, p.y); mn.z = std::min(mn.z, p.z);
        mx.x = std::max(mx.x, p.x); mx.y = std::max(mx.y, p.y); mx.z = std::max(mx.z, p.z);
    }
    void expand(const Box& b) { expand_min(b.mn); expand_max(b.mx); }      // Aabb.h:28-37
    void expand_min(V3 p) { mn.x = std::min(mn.x, p.x); mn.y = std::min(mn.y, p.y); mn.z = std::min(mn.z, p.z); }
    void expand_max(V3 p) { mx.x = std::max(mx.x, p.x); mx.y = std::max(mx.y, p.y); mx.z = std::max(mx.z, p.z); }
};

void mat_identity(float m[16]) {
    for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}
// Mat4f.h:38-49
void mat_mul(float out[16], const float a[16], const float b[16]) {
    float r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0;
            for (int k = 0; k < 4; k++) acc += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = acc;
        }
    std::memcpy(out, r, sizeof(r));
}
// Mat4f.h:61-123 (Gauss-Jordan, partial pivot, singular -> identity)
void mat_inverse(float out[16], const float m[16]) {
    float t[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            t[i][j] = m[i * 4 + j];
            t[i][j + 4] = (i == j) ? 1.0f : 0.0f;
        }
    for (int i = 0; i < 4; i++) {
        int pivot = i;
        for (int j = i + 1; j < 4; j++)
            if (std::fabs(t[j][i]) > std::fabs(t[pivot][i])) pivot = j;
        if (pivot != i)
            for (int j = 0; j < 8; j++) std::swap(t[i][j], t[pivot][j]);
        float scale = t[i][i];
        if (std::fabs(scale) < 1e-8f) { mat_identity(out); return; }
        for (int j = 0; j < 8; j++) t[i][j] /= scale;
        for (int j = 0; j < 4; j++)
            if (i != j) {
                float factor = t[j][i];
                for (int k = 0; k < 8; k++) t[j][k] -= factor * t[i][k];
            }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = t[i][j + 4];
}
// Mat4f.h:21-27
V3 xf_point(const float* m, V3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1;
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1;
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1;
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * 1;
    return {x / w, y / w, z / w};
}
// parser.cpp:113-135
Box transform_box(const Box& b, const float* m) {
    V3 c[8] = {{b.mn.x, b.mn.y, b.mn.z}, {b.mx.x, b.mn.y, b.mn.z}, {b.mn.x, b.mx.y, b.mn.z}, {b.mx.x, b.mx.y, b.mn.z},
               {b.mn.x, b.mn.y, b.mx.z}, {b.mx.x, b.mn.y, b.mx.z}, {b.mn.x, b.mx.y, b.mx.z}, {b.mx.x, b.mx.y, b.mx.z}};
    Box r;
    for (int i = 0; i < 8; i++) r.expand(xf_point(m, c[i]));
    return r;
}

// ---- midpoint-split BVH, restating raytracer.cpp:151-204 / 283-339 ----
struct BuildPrim { Box box; V3 centroid; };

// Builds into `nodes` (node 0 = root) and permutes `idx`.  Returns nodesUsed.
uint32_t build_bvh(const std::vector<BuildPrim>& prims, std::vector<uint32_t>& idx, std::vector<rt_bvh_node>& nodes) {
    const uint32_t N = (uint32_t)prims.size();
    idx.resize(N);
    for (uint32_t i = 0; i < N; i++) idx[i] = i;
    nodes.assign(N ? 2 * (size_t)N : 1, rt_bvh_node{});
    uint32_t used = 1;
    auto update_bounds = [&](uint32_t n) {
        Box b;
        for (int32_t i = 0; i < nodes[n].prim_count; i++) b.expand(prims[idx[nodes[n].left_or_first + i]].box);
        st(nodes[n].bmin, b.mn);
        st(nodes[n].bmax, b.mx);
    };
    nodes[0].left_or_first = 0;
    nodes[0].prim_count = (int32_t)N;
    update_bounds(0);
    // The reference recurses left then right, allocating both children before
    // descending; an explicit stack that pushes right then left numbers the
    // nodes identically.
    std::vector<uint32_t> work{0};
    while (!work.empty()) {
        uint32_t n = work.back();
        work.pop_back();
        if (nodes[n].prim_count <= 2) continue;
        float ext[3] = {nodes[n].bmax[0] - nodes[n].bmin[0], nodes[n].bmax[1] - nodes[n].bmin[1],
                        nodes[n].bmax[2] - nodes[n].bmin[2]};
        int axis = 0;
        if (ext[1] > ext[0]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        float split = nodes[n].bmin[axis] + ext[axis] * 0.5f;
        int i = nodes[n].left_or_first;
        int j = i + nodes[n].prim_count - 1;
        while (i <= j) {
            const V3& c = prims[idx[i]].centroid;
            float cv = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
            if (cv < split) i++;
            else std::swap(idx[i], idx[j--]);
        }
        int left_count = i - nodes[n].left_or_first;
        if (left_count == 0 || left_count == nodes[n].prim_count) continue;
        uint32_t l = used++, r = used++;
        nodes[l].left_or_first = nodes[n].left_or_first;
        nodes[l].prim_count = left_count;
        nodes[r].left_or_first = i;
        nodes[r].prim_count = nodes[n].prim_count - left_count;
        nodes[n].left_or_first = (int32_t)l;
        nodes[n].prim_count = 0;
        update_bounds(l);
        update_bounds(r);
        work.push_back(r);
        work.push_back(l);
    }
    nodes.resize(used);
    return used;
}

}  // namespace

struct rth_scene { RtSceneStore st; };

struct rth_builder {
    RtSceneStore st;
    struct MeshExtra { Box local, world; bool is_instance; int original; std::vector<rt_face> faces; };
    std::vector<MeshExtra> mesh_extra;
    std::vector<Box> tri_local, tri_world, sph_local, sph_world;
    std::vector<int> touched;
    std::unordered_set<int> touched_set;
};

// include/rt_host.h: nothing throws across the C boundary (ctypes callers would abort): host allocation failures
// and any other exception become a NULL / negative return with a message
#define RTH_GUARD_BEGIN try {
#define RTH_GUARD_END(ret) } catch (const std::bad_alloc&) { g_err = "host allocation failed"; return ret; } \
    catch (const std::exception& e_) { g_err = std::string("internal error: ") + e_.what(); return ret; } catch (...) { g_err = "internal error"; return ret; }

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

rth_scene* rth_scene_load(const char* path) {
    RTH_GUARD_BEGIN
    rth_scene* s = new rth_scene();
    if (!s->st.load(path)) {
        g_err = std::string("cannot load scene blob ") + (path ? path : "(null)");
        delete s;
        return nullptr;
    }
    return s;
    RTH_GUARD_END(nullptr)
}
int rth_scene_save(const rth_scene* s, const char* path) {
    RTH_GUARD_BEGIN
    if (!s || !s->st.save(path)) { g_err = "cannot write scene blob"; return RT_ERR_IO; }
    return RT_OK;
    RTH_GUARD_END(RT_ERR_IO)
}
const rt_scene_desc* rth_scene_desc(const rth_scene* s) { return s ? &s->st.desc : nullptr; }
int rth_scene_num_cameras(const rth_scene* s) { return s ? (int)s->st.cameras.size() : 0; }
const rt_camera_desc* rth_scene_camera(const rth_scene* s, int i) {
    if (!s || i < 0 || i >= (int)s->st.cameras.size()) return nullptr;
    return &s->st.cameras[i];
}
const char* rth_scene_camera_name(const rth_scene* s, int i) {
    if (!s || i < 0 || i >= (int)s->st.camera_names.size()) return "";
    return s->st.camera_names[i].c_str();
}
int rth_scene_set_camera_resolution(rth_scene* s, int i, int width, int height, int num_samples) {
    if (!s || i < 0 || i >= (int)s->st.cameras.size() || width < 1 || height < 1 || num_samples < 1) return RT_ERR_INVALID;
    rt_camera_desc& c = s->st.cameras[i];
    // pixel_width = (r-l)/image_width (parser.cpp:438-439): rescale from the old extent
    float extent_w = c.pixel_width * c.image_width, extent_h = c.pixel_height * c.image_height;
    c.image_width = width; c.image_height = height;
    c.pixel_width = extent_w / width; c.pixel_height = extent_h / height;
    c.num_samples = num_samples;
    c.samples_per_side = (int)std::sqrt((double)num_samples);   // parser.cpp:411
    return RT_OK;
}
int rth_scene_set_max_depth(rth_scene* s, int d) {
    if (!s) return RT_ERR_INVALID;
    s->st.max_recursion_depth = d;
    s->st.finalize();
    return RT_OK;
}
void rth_scene_free(rth_scene* s) { delete s; }

// ------------------------------------------------------------------ builder
rth_builder* rth_builder_new(void) { RTH_GUARD_BEGIN return new rth_builder(); RTH_GUARD_END(nullptr) }
void rth_builder_free(rth_builder* b) { delete b; }

void rth_builder_set_globals(rth_builder* b, const float bg[3], const float amb[3], float eps, int depth) {
    std::memcpy(b->st.background_color, bg, 12);
    std::memcpy(b->st.ambient_light, amb, 12);
    b->st.shadow_ray_epsilon = eps;
    b->st.max_recursion_depth = depth;
}
int rth_builder_add_vertices(rth_builder* b, const float* xyz, int n) {
    RTH_GUARD_BEGIN
    int first = (int)b->st.vertex_data.size() + 1;
    for (int i = 0; i < n; i++) {
        rt_vertex v{};
        v.pos[0] = xyz[3 * i]; v.pos[1] = xyz[3 * i + 1]; v.pos[2] = xyz[3 * i + 2];
        b->st.vertex_data.push_back(v);
    }
    return first;
    RTH_GUARD_END(RT_ERR_OOM)
}
int rth_builder_add_material(rth_builder* b, const rt_material* m) {
    RTH_GUARD_BEGIN
    b->st.materials.push_back(*m);
    return (int)b->st.materials.size();
    RTH_GUARD_END(RT_ERR_OOM)
}
void rth_builder_add_point_light(rth_builder* b, const float pos[3], const float inten[3]) {
    rt_point_light l;
    std::memcpy(l.position, pos, 12);
    std::memcpy(l.intensity, inten, 12);
    b->st.point_lights.push_back(l);
}
void rth_builder_add_area_light(rth_builder* b, const float pos[3], const float normal[3], float size, const float rad[3]) {
    rt_area_light l;
    std::memcpy(l.position, pos, 12);
    st(l.normal, normalize(ld(normal)));   // parser.cpp:320
    l.size = size;
    std::memcpy(l.radiance, rad, 12);
    b->st.area_lights.push_back(l);
}

static rt_face make_face(const std::vector<rt_vertex>& vd, int i0, int i1, int i2, V3* face_norm_out) {
    rt_face f{};
    f.i0 = i0; f.i1 = i1; f.i2 = i2;
    V3 va = ld(vd[i0 - 1].pos), vb = ld(vd[i1 - 1].pos), vc = ld(vd[i2 - 1].pos);
    V3 fn = cross(vb - va, vc - va);       // parser.cpp:621
    V3 nu = normalize(fn);                 // parser.cpp:622
    st(f.n_unit, nu);
    f.plane_d = -(dot(nu, va));            // parser.cpp:623
    if (face_norm_out) *face_norm_out = fn;
    return f;
}

static void set_transform(uint32_t& flags, float M[16], float Minv[16], float mb[3], const float* transform,
                          const float* motion_blur) {
    if (transform) {
        std::memcpy(M, transform, 64);
        mat_inverse(Minv, M);
        flags |= RT_OBJ_HAS_TRANSFORM;
    } else {
        mat_identity(M);
        mat_identity(Minv);
    }
    mb[0] = mb[1] = mb[2] = 0;
    if (motion_blur) { std::memcpy(mb, motion_blur, 12); flags |= RT_OBJ_HAS_MOTION_BLUR; }
}

int rth_builder_add_mesh(rth_builder* b, const int32_t* faces, int n_faces, int material_id, int smooth,
                         const float* transform, const float* motion_blur) {
    RTH_GUARD_BEGIN
    const int nv = (int)b->st.vertex_data.size();
    rth_builder::MeshExtra ex;
    ex.is_instance = false;
    ex.original = -1;
    ex.faces.reserve(n_faces);
    for (int k = 0; k < n_faces; k++) {
        int i0 = faces[3 * k], i1 = faces[3 * k + 1], i2 = faces[3 * k + 2];
        if (i0 < 1 || i1 < 1 || i2 < 1 || i0 > nv || i1 > nv || i2 > nv) { g_err = "face index out of range"; return RT_ERR_INVALID; }
        V3 fn;
        rt_face f = make_face(b->st.vertex_data, i0, i1, i2, &fn);
        ex.local.expand(ld(b->st.vertex_data[i0 - 1].pos));
        ex.local.expand(ld(b->st.vertex_data[i1 - 1].pos));
        ex.local.expand(ld(b->st.vertex_data[i2 - 1].pos));
        if (smooth) {   // parser.cpp:626-634
            for (int id : {i0, i1, i2}) {
                rt_vertex& v = b->st.vertex_data[id - 1];
                st(v.normal, ld(v.normal) + fn);
                if (b->touched_set.insert(id).second) b->touched.push_back(id);
            }
        }
        ex.faces.push_back(f);
    }
    rt_mesh m{};
    m.bvh_index = -1;
    m.material_id = material_id;
    m.flags = smooth ? RT_OBJ_SMOOTH : 0u;
    set_transform(m.flags, m.transform, m.inv_transform, m.motion_blur, transform, motion_blur);
    ex.world = transform ? transform_box(ex.local, m.transform) : ex.local;   // parser.cpp:644-649
    b->st.meshes.push_back(m);
    b->mesh_extra.push_back(std::move(ex));
    return (int)b->st.meshes.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

int rth_builder_add_instance(rth_builder* b, int base, int material_id, const float* transform, int reset,
                             const float* motion_blur) {
    RTH_GUARD_BEGIN
    if (base < 0 || base >= (int)b->st.meshes.size()) { g_err = "bad base mesh"; return RT_ERR_INVALID; }
    int orig = base;
    while (b->mesh_extra[orig].is_instance) orig = b->mesh_extra[orig].original;   // parser.cpp:694-698
    float inst[16];
    if (transform) std::memcpy(inst, transform, 64); else mat_identity(inst);
    float fin[16];
    if (reset) std::memcpy(fin, inst, 64); else mat_mul(fin, inst, b->st.meshes[base].transform);   // parser.cpp:716-721
    rt_mesh m{};
    m.bvh_index = -1;
    m.material_id = material_id > 0 ? material_id : b->st.meshes[base].material_id;
    m.flags = (b->st.meshes[orig].flags & RT_OBJ_SMOOTH);
    set_transform(m.flags, m.transform, m.inv_transform, m.motion_blur, fin, motion_blur);   // hasTransform = true (parser.cpp:730)
    rth_builder::MeshExtra ex;
    ex.is_instance = true;
    ex.original = orig;
    ex.local = b->mesh_extra[orig].local;
    ex.world = transform_box(ex.local, m.transform);   // parser.cpp:735
    b->st.meshes.push_back(m);
    b->mesh_extra.push_back(std::move(ex));
    return (int)b->st.meshes.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

int rth_builder_add_triangle(rth_builder* b, int i0, int i1, int i2, int material_id, const float* transform,
                             const float* motion_blur) {
    RTH_GUARD_BEGIN
    const int nv = (int)b->st.vertex_data.size();
    if (i0 < 1 || i1 < 1 || i2 < 1 || i0 > nv || i1 > nv || i2 > nv) { g_err = "triangle index out of range"; return RT_ERR_INVALID; }
    rt_triangle t{};
    t.material_id = material_id;
    t.face = make_face(b->st.vertex_data, i0, i1, i2, nullptr);
    set_transform(t.flags, t.transform, t.inv_transform, t.motion_blur, transform, motion_blur);
    Box local;
    local.expand(ld(b->st.vertex_data[i0 - 1].pos));
    local.expand(ld(b->st.vertex_data[i1 - 1].pos));
    local.expand(ld(b->st.vertex_data[i2 - 1].pos));
    b->tri_local.push_back(local);
    b->tri_world.push_back(transform ? transform_box(local, t.transform) : local);
    b->st.triangles.push_back(t);
    return (int)b->st.triangles.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

int rth_builder_add_sphere(rth_builder* b, int center_id, float radius, int material_id, const float* transform,
                           const float* motion_blur) {
    RTH_GUARD_BEGIN
    if (center_id < 1 || center_id > (int)b->st.vertex_data.size()) { g_err = "sphere centre out of range"; return RT_ERR_INVALID; }
    rt_sphere s{};
    s.material_id = material_id;
    s.center_vertex_id = center_id;
    s.radius = radius;
    set_transform(s.flags, s.transform, s.inv_transform, s.motion_blur, transform, motion_blur);
    V3 c = ld(b->st.vertex_data[center_id - 1].pos), r{radius, radius, radius};
    Box local;
    local.mn = c - r;     // parser.cpp:836-839
    local.mx = c + r;
    b->sph_local.push_back(local);
    b->sph_world.push_back(transform ? transform_box(local, s.transform) : local);
    b->st.spheres.push_back(s);
    return (int)b->st.spheres.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

int rth_builder_add_plane(rth_builder* b, int point_id, const float normal[3], int material_id) {
    RTH_GUARD_BEGIN
    if (point_id < 1 || point_id > (int)b->st.vertex_data.size()) { g_err = "plane point out of range"; return RT_ERR_INVALID; }
    rt_plane p{};
    p.material_id = material_id;
    V3 n = normalize(ld(normal));   // parser.cpp:885-889
    st(p.n_unit, n);
    p.plane_d = -(dot(n, ld(b->st.vertex_data[point_id - 1].pos)));
    b->st.planes.push_back(p);
    return (int)b->st.planes.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

int rth_builder_add_camera_lookat(rth_builder* b, const float position[3], const float gaze_point[3], const float up[3],
                                  float fovy_deg, float near_distance, int width, int height, int num_samples,
                                  float aperture_size, float focus_distance, const char* name) {
    RTH_GUARD_BEGIN
    // parser.cpp:343-446 (lookAt branch, no camera Transformations)
    V3 pos = ld(position);
    V3 gaze = ld(gaze_point) - pos;
    V3 w = normalize(gaze * -1);
    V3 vprime = normalize(ld(up));
    V3 u = normalize(cross(vprime, w));
    V3 v = cross(w, u);
    const float PI = 3.14159265358979323846f;
    float thetaY = fovy_deg * float(PI) / 180.0f;           // MathF.h:9
    float t = near_distance * std::tan(thetaY * 0.5f);
    float bb = -t;
    float aspect = (height > 0) ? (float(width) / float(height)) : 1.0f;
    float r = aspect * t;
    float l = -r;
    rt_camera_desc c{};
    st(c.position, pos); st(c.u, u); st(c.v, v); st(c.w, w);
    c.image_width = width; c.image_height = height;
    c.num_samples = num_samples < 1 ? 1 : num_samples;
    c.samples_per_side = (int)std::sqrt((double)c.num_samples);
    c.aperture_size = aperture_size > 0 ? aperture_size : 0.0f;
    c.focus_distance = focus_distance;
    c.has_depth_of_field = aperture_size > 0.0f ? 1 : 0;
    c.pixel_width = (r - l) / width;
    c.pixel_height = (t - bb) / height;
    V3 m = pos - w * near_distance;
    V3 q = m + u * l + v * t;
    st(c.q, q);
    b->st.cameras.push_back(c);
    b->st.camera_names.push_back(name ? name : "out.png");
    return (int)b->st.cameras.size() - 1;
    RTH_GUARD_END(RT_ERR_OOM)
}

rth_scene* rth_builder_finish(rth_builder* b) {
    RTH_GUARD_BEGIN
    rth_scene* out = new rth_scene();
    RtSceneStore& s = out->st;
    s = b->st;
    // parser.cpp:670-673: normalise accumulated smooth normals (on the copy)
    for (int id : b->touched) st(s.vertex_data[id - 1].normal, normalize(ld(s.vertex_data[id - 1].normal)));

    // BuildAllMeshBVHs, raytracer.cpp:208-231
    s.mesh_bvhs.clear(); s.mesh_nodes.clear(); s.mesh_prim_face.clear(); s.faces.clear();
    for (size_t k = 0; k < s.meshes.size(); k++) {
        const auto& ex = b->mesh_extra[k];
        if (ex.is_instance) { s.meshes[k].bvh_index = s.meshes[ex.original].bvh_index; continue; }
        s.meshes[k].bvh_index = (int32_t)s.mesh_bvhs.size();
        rt_mesh_bvh mb{};
        mb.node_offset = (uint32_t)s.mesh_nodes.size();
        mb.prim_offset = (uint32_t)s.mesh_prim_face.size();
        mb.face_offset = (uint32_t)s.faces.size();
        const uint32_t N = (uint32_t)ex.faces.size();
        if (N > 0) {
            std::vector<BuildPrim> prims(N);   // MakeMeshPrimsArray, raytracer.cpp:259-281
            for (uint32_t i = 0; i < N; i++) {
                const rt_face& f = ex.faces[i];
                V3 v0 = ld(s.vertex_data[f.i0 - 1].pos), v1 = ld(s.vertex_data[f.i1 - 1].pos), v2 = ld(s.vertex_data[f.i2 - 1].pos);
                prims[i].box.expand(v0); prims[i].box.expand(v1); prims[i].box.expand(v2);
                prims[i].centroid = (v0 + v1 + v2) * (1.0f / 3.0f);
            }
            std::vector<uint32_t> idx;
            std::vector<rt_bvh_node> nodes;
            mb.node_count = build_bvh(prims, idx, nodes);
            mb.prim_count = N;
            mb.face_count = N;
            s.mesh_nodes.insert(s.mesh_nodes.end(), nodes.begin(), nodes.end());
            s.mesh_prim_face.insert(s.mesh_prim_face.end(), idx.begin(), idx.end());   // prims[i].faceIdx == i
            s.faces.insert(s.faces.end(), ex.faces.begin(), ex.faces.end());
        }
        s.mesh_bvhs.push_back(mb);
    }

    // BuildTopLevelBVH, raytracer.cpp:98-149: meshes, then spheres, then triangles
    std::vector<BuildPrim> tp;
    s.top_prims.clear();
    auto add_top = [&](uint32_t kind, uint32_t index, const Box& box) {
        BuildPrim p;
        p.box = box;
        p.centroid = (box.mn + box.mx) * 0.5f;
        tp.push_back(p);
        s.top_prims.push_back(rt_top_prim{kind, index});
    };
    for (size_t k = 0; k < s.meshes.size(); k++)
        add_top(RT_KIND_MESH, (uint32_t)k, (s.meshes[k].flags & RT_OBJ_HAS_TRANSFORM) ? b->mesh_extra[k].world : b->mesh_extra[k].local);
    for (size_t k = 0; k < s.spheres.size(); k++)
        add_top(RT_KIND_SPHERE, (uint32_t)k, (s.spheres[k].flags & RT_OBJ_HAS_TRANSFORM) ? b->sph_world[k] : b->sph_local[k]);
    for (size_t k = 0; k < s.triangles.size(); k++)
        add_top(RT_KIND_TRIANGLE, (uint32_t)k, (s.triangles[k].flags & RT_OBJ_HAS_TRANSFORM) ? b->tri_world[k] : b->tri_local[k]);
    if (!tp.empty()) build_bvh(tp, s.top_prim_idx, s.top_nodes);
    else { s.top_nodes.clear(); s.top_prim_idx.clear(); }
    s.finalize();
    return out;
    RTH_GUARD_END(nullptr)
}

void rth_mat_translation(float out[16], float x, float y, float z) { mat_identity(out); out[3] = x; out[7] = y; out[11] = z; }
void rth_mat_scaling(float out[16], float x, float y, float z) { mat_identity(out); out[0] = x; out[5] = y; out[10] = z; }
// parser.cpp:48-70
void rth_mat_rotation(float out[16], float angle_deg, float ax, float ay, float az) {
    float rad = angle_deg * 3.14159265358979323846 / 180.0f;
    float c = cosf(rad), s = sinf(rad), t = 1.0f - c;
    V3 a = normalize(V3{ax, ay, az});
    mat_identity(out);
    out[0] = t * a.x * a.x + c;        out[1] = t * a.x * a.y - s * a.z;  out[2] = t * a.x * a.z + s * a.y;
    out[4] = t * a.x * a.y + s * a.z;  out[5] = t * a.y * a.y + c;        out[6] = t * a.y * a.z - s * a.x;
    out[8] = t * a.x * a.z - s * a.y;  out[9] = t * a.y * a.z + s * a.x;  out[10] = t * a.z * a.z + c;
}
void rth_mat_mul(float out[16], const float a[16], const float b[16]) { mat_mul(out, a, b); }

}  // extern "C"
