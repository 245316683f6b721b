// rt_flatten.hpp — reference-side binding: flatten the reference's `Scene`
// (parser.h:192-215) and its BVH globals (raytracer.cpp:8-11 topPrims /
// topPrimIdx / topBvhNodes / rootNodeIdx / nodesUsed, Bvh.h:76 meshBVHs) into
// the POD arrays of rt_scene_desc (include/rt.h).
//
// This header is written against the reference's own types: include it AFTER
// "raytracer.h" in the translation unit that defines those globals (the
// reference puts non-inline functions and a global into its headers —
// Bvh.h:17-37,76, HitRecord.h:15-23 — so there can only be one such unit).
// It copies values only; no arithmetic is performed on scene data.
#pragma once
#include <cstring>
#include "rt_scene_store.hpp"

static inline void rt_copy_mat(float dst[16], const Mat4f& m) { std::memcpy(dst, m.m, 16 * sizeof(float)); }
static inline void rt_copy_v3(float dst[3], const Vec3f& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

static inline rt_bvh_node rt_copy_node(const BVHNode& n) {
    rt_bvh_node o;
    rt_copy_v3(o.bmin, n.bounds.min);
    rt_copy_v3(o.bmax, n.bounds.max);
    o.left_or_first = n.leftNodeIdx;   // union with firstPrimIdx (Bvh.h:42-45)
    o.prim_count    = n.primCount;
    return o;
}

static inline rt_face rt_copy_face(const Face& f) {
    rt_face o;
    o.i0 = f.i0; o.i1 = f.i1; o.i2 = f.i2;
    rt_copy_v3(o.n_unit, f.n_unit);
    o.plane_d = f.plane_d;
    return o;
}

static inline rt_camera_desc rt_flatten_camera(const Camera& c) {
    rt_camera_desc o;
    std::memset(&o, 0, sizeof(o));
    rt_copy_v3(o.position, c.position);
    rt_copy_v3(o.u, c.u); rt_copy_v3(o.v, c.v); rt_copy_v3(o.w, c.w);
    rt_copy_v3(o.q, c.q);
    o.pixel_width = c.pixel_width;   o.pixel_height = c.pixel_height;
    o.image_width = c.image_width;   o.image_height = c.image_height;
    o.num_samples = c.num_samples;   o.samples_per_side = c.samples_per_side;
    o.aperture_size = c.aperture_size;
    o.focus_distance = c.focus_distance;
    o.has_depth_of_field = c.has_depth_of_field ? 1 : 0;
    return o;
}

// Call after BuildAllMeshBVHs(scene); BuildTopLevelBVH(scene) (raytracer.cpp:25-26).
static inline bool rt_flatten(const Scene& scene, RtSceneStore& out) {
    if (rootNodeIdx != 0) return false;   // the reference never changes it (raytracer.cpp:11)
    out = RtSceneStore();
    rt_copy_v3(out.background_color, scene.background_color);
    rt_copy_v3(out.ambient_light, scene.ambient_light);
    out.shadow_ray_epsilon  = scene.shadow_ray_epsilon;
    out.max_recursion_depth = scene.max_recursion_depth;

    // top level
    for (uint32_t i = 0; i < nodesUsed && i < topBvhNodes.size(); i++) out.top_nodes.push_back(rt_copy_node(topBvhNodes[i]));
    out.top_prim_idx.assign(topPrimIdx.begin(), topPrimIdx.end());
    for (const TopPrim& p : topPrims) out.top_prims.push_back(rt_top_prim{(uint32_t)p.kind, p.index});

    // mesh BVHs; the face array of BVH b is that of the non-instance mesh that owns it
    std::vector<int> owner(meshBVHs.size(), -1);
    for (size_t k = 0; k < scene.meshes.size(); k++) {
        const Mesh& m = scene.meshes[k];
        if (!m.isInstance && m.bvhIndex >= 0 && owner[m.bvhIndex] < 0) owner[m.bvhIndex] = (int)k;
    }
    for (size_t b = 0; b < meshBVHs.size(); b++) {
        const MeshBVH& bvh = meshBVHs[b];
        rt_mesh_bvh mb;
        mb.node_offset = (uint32_t)out.mesh_nodes.size();
        mb.prim_offset = (uint32_t)out.mesh_prim_face.size();
        mb.face_offset = (uint32_t)out.faces.size();
        const uint32_t N = (uint32_t)bvh.prims.size();
        if (N == 0) {          // BuildMeshBVH returned early (raytracer.cpp:241): nothing to hit
            mb.node_count = 0; mb.prim_count = 0; mb.face_count = 0;
            out.mesh_bvhs.push_back(mb);
            continue;
        }
        mb.node_count = bvh.nodesUsed;
        mb.prim_count = N;
        for (uint32_t i = 0; i < bvh.nodesUsed; i++) out.mesh_nodes.push_back(rt_copy_node(bvh.nodes[i]));
        for (uint32_t p = 0; p < N; p++) out.mesh_prim_face.push_back(bvh.prims[bvh.primIdx[p]].faceIdx);
        const Mesh& om = scene.meshes[owner[b]];
        mb.face_count = (uint32_t)om.faces.size();
        for (const Face& f : om.faces) out.faces.push_back(rt_copy_face(f));
        out.mesh_bvhs.push_back(mb);
    }

    for (const Vertex& v : scene.vertex_data) {
        rt_vertex o; rt_copy_v3(o.pos, v.pos); rt_copy_v3(o.normal, v.normal);
        out.vertex_data.push_back(o);
    }
    for (const Mesh& m : scene.meshes) {
        rt_mesh o; std::memset(&o, 0, sizeof(o));
        o.bvh_index = m.bvhIndex;
        o.material_id = m.material_id;
        o.flags = (m.hasTransform ? RT_OBJ_HAS_TRANSFORM : 0u) | (m.has_motion_blur ? RT_OBJ_HAS_MOTION_BLUR : 0u) |
                  (m.is_smooth ? RT_OBJ_SMOOTH : 0u);
        rt_copy_mat(o.transform, m.transformation);
        rt_copy_mat(o.inv_transform, m.invTransformation);
        if (m.has_motion_blur) rt_copy_v3(o.motion_blur, m.motion_blur);
        out.meshes.push_back(o);
    }
    for (const Triangle& t : scene.triangles) {
        rt_triangle o; std::memset(&o, 0, sizeof(o));
        o.material_id = t.material_id;
        o.face = rt_copy_face(t.face);
        o.flags = (t.hasTransform ? RT_OBJ_HAS_TRANSFORM : 0u) | (t.has_motion_blur ? RT_OBJ_HAS_MOTION_BLUR : 0u);
        rt_copy_mat(o.transform, t.transformation);
        rt_copy_mat(o.inv_transform, t.invTransformation);
        if (t.has_motion_blur) rt_copy_v3(o.motion_blur, t.motion_blur);
        out.triangles.push_back(o);
    }
    for (const Sphere& s : scene.spheres) {
        rt_sphere o; std::memset(&o, 0, sizeof(o));
        o.material_id = s.material_id;
        o.center_vertex_id = s.center_vertex_id;
        o.radius = s.radius;
        o.flags = (s.hasTransform ? RT_OBJ_HAS_TRANSFORM : 0u) | (s.has_motion_blur ? RT_OBJ_HAS_MOTION_BLUR : 0u);
        rt_copy_mat(o.transform, s.transformation);
        rt_copy_mat(o.inv_transform, s.invTransformation);
        if (s.has_motion_blur) rt_copy_v3(o.motion_blur, s.motion_blur);
        out.spheres.push_back(o);
    }
    for (const Plane& p : scene.planes) {
        rt_plane o; o.material_id = p.material_id; rt_copy_v3(o.n_unit, p.n_unit); o.plane_d = p.plane_d;
        out.planes.push_back(o);
    }
    for (const Material& m : scene.materials) {
        rt_material o; std::memset(&o, 0, sizeof(o));
        o.type = (uint32_t)m.type;
        rt_copy_v3(o.ambient_refl, m.ambient_refl);   rt_copy_v3(o.diffuse_refl, m.diffuse_refl);
        rt_copy_v3(o.specular_refl, m.specular_refl); rt_copy_v3(o.mirror_refl, m.mirror_refl);
        o.phong_exponent = m.phong_exponent;     o.refraction_index = m.refraction_index;
        o.absorption_index = m.absorption_index; rt_copy_v3(o.absorption_coef, m.absorption_coef);
        o.roughness = m.roughness;
        out.materials.push_back(o);
    }
    for (const PointLight& l : scene.point_lights) {
        rt_point_light o; rt_copy_v3(o.position, l.position); rt_copy_v3(o.intensity, l.intensity);
        out.point_lights.push_back(o);
    }
    for (const AreaLight& l : scene.area_lights) {
        rt_area_light o; rt_copy_v3(o.position, l.position); rt_copy_v3(o.normal, l.normal);
        o.size = l.size; rt_copy_v3(o.radiance, l.radiance);
        out.area_lights.push_back(o);
    }
    for (const Camera& c : scene.cameras) {
        out.cameras.push_back(rt_flatten_camera(c));
        out.camera_names.push_back(c.image_name);
    }
    out.finalize();
    return true;
}
