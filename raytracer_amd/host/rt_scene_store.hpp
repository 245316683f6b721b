// rt_scene_store.hpp — owning container for a flattened scene (rt_scene_desc)
// plus its cameras, with a trivial binary (de)serialiser ("RTSCENE1" blobs).
//
// Host-side plumbing only: the arrays are exactly the ones rt.h documents
// (reference Scene + BVH globals, parser.h:192-207, raytracer.cpp:8-11,
// Bvh.h:76), kept in std::vectors so that rt_scene_desc can point into them.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rt.h"

struct RtSceneStore {
    float   background_color[3] = {0, 0, 0};
    float   ambient_light[3]    = {0, 0, 0};
    float   shadow_ray_epsilon  = 1e-3f;   // parser.cpp:162
    int32_t max_recursion_depth = 6;       // parser.cpp:169

    std::vector<rt_bvh_node>    top_nodes;
    std::vector<uint32_t>       top_prim_idx;
    std::vector<rt_top_prim>    top_prims;
    std::vector<rt_mesh_bvh>    mesh_bvhs;
    std::vector<rt_bvh_node>    mesh_nodes;
    std::vector<uint32_t>       mesh_prim_face;
    std::vector<rt_face>        faces;
    std::vector<rt_vertex>      vertex_data;
    std::vector<rt_mesh>        meshes;
    std::vector<rt_triangle>    triangles;
    std::vector<rt_sphere>      spheres;
    std::vector<rt_plane>       planes;
    std::vector<rt_material>    materials;
    std::vector<rt_point_light> point_lights;
    std::vector<rt_area_light>  area_lights;

    std::vector<rt_camera_desc> cameras;
    std::vector<std::string>    camera_names;   // Camera::image_name

    rt_scene_desc desc;

    // Point desc at the vectors.  Call after any mutation.
    void finalize() {
        std::memset(&desc, 0, sizeof(desc));
        std::memcpy(desc.background_color, background_color, sizeof(background_color));
        std::memcpy(desc.ambient_light, ambient_light, sizeof(ambient_light));
        desc.shadow_ray_epsilon  = shadow_ray_epsilon;
        desc.max_recursion_depth = max_recursion_depth;
        desc.top_nodes = top_nodes.data();           desc.n_top_nodes = (uint32_t)top_nodes.size();
        desc.top_prim_idx = top_prim_idx.data();     desc.n_top_prims = (uint32_t)top_prims.size();
        desc.top_prims = top_prims.data();
        desc.mesh_bvhs = mesh_bvhs.data();           desc.n_mesh_bvhs = (uint32_t)mesh_bvhs.size();
        desc.mesh_nodes = mesh_nodes.data();         desc.n_mesh_nodes = (uint32_t)mesh_nodes.size();
        desc.mesh_prim_face = mesh_prim_face.data(); desc.n_mesh_prims = (uint32_t)mesh_prim_face.size();
        desc.faces = faces.data();                   desc.n_faces = (uint32_t)faces.size();
        desc.vertex_data = vertex_data.data();       desc.n_vertices = (uint32_t)vertex_data.size();
        desc.meshes = meshes.data();                 desc.n_meshes = (uint32_t)meshes.size();
        desc.triangles = triangles.data();           desc.n_triangles = (uint32_t)triangles.size();
        desc.spheres = spheres.data();               desc.n_spheres = (uint32_t)spheres.size();
        desc.planes = planes.data();                 desc.n_planes = (uint32_t)planes.size();
        desc.materials = materials.data();           desc.n_materials = (uint32_t)materials.size();
        desc.point_lights = point_lights.data();     desc.n_point_lights = (uint32_t)point_lights.size();
        desc.area_lights = area_lights.data();       desc.n_area_lights = (uint32_t)area_lights.size();
    }

    // ---- blob I/O -----------------------------------------------------
    template <class T> static bool wr(FILE* f, const std::vector<T>& v) {
        uint64_t n = v.size();
        if (fwrite(&n, 8, 1, f) != 1) return false;
        return n == 0 || fwrite(v.data(), sizeof(T), n, f) == n;
    }
    // A count comes from the file: it is believed only as far as the file has bytes left for it (a damaged or
    // hostile blob must not make the loader allocate gigabytes).
    template <class T> static bool rd(FILE* f, std::vector<T>& v) {
        uint64_t n = 0;
        if (fread(&n, 8, 1, f) != 1) return false;
        const long here = ftell(f);
        if (here < 0 || fseek(f, 0, SEEK_END) != 0) return false;
        const long end = ftell(f);
        if (end < here || fseek(f, here, SEEK_SET) != 0) return false;
        if (n > (uint64_t)(end - here) / sizeof(T)) return false;
        v.resize(n);
        return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
    }

    bool save(const char* path) const {
        FILE* f = fopen(path, "wb");
        if (!f) return false;
        bool ok = fwrite("RTSCENE1", 8, 1, f) == 1;
        uint32_t sizes[12] = {sizeof(rt_bvh_node), sizeof(rt_top_prim), sizeof(rt_mesh_bvh), sizeof(rt_face),
                              sizeof(rt_vertex), sizeof(rt_mesh), sizeof(rt_triangle), sizeof(rt_sphere),
                              sizeof(rt_plane), sizeof(rt_material), sizeof(rt_point_light), sizeof(rt_camera_desc)};
        ok = ok && fwrite(sizes, sizeof(sizes), 1, f) == 1;
        ok = ok && fwrite(background_color, 12, 1, f) == 1 && fwrite(ambient_light, 12, 1, f) == 1;
        ok = ok && fwrite(&shadow_ray_epsilon, 4, 1, f) == 1 && fwrite(&max_recursion_depth, 4, 1, f) == 1;
        ok = ok && wr(f, top_nodes) && wr(f, top_prim_idx) && wr(f, top_prims) && wr(f, mesh_bvhs) &&
             wr(f, mesh_nodes) && wr(f, mesh_prim_face) && wr(f, faces) && wr(f, vertex_data) && wr(f, meshes) &&
             wr(f, triangles) && wr(f, spheres) && wr(f, planes) && wr(f, materials) && wr(f, point_lights) &&
             wr(f, area_lights) && wr(f, cameras);
        uint64_t nn = camera_names.size();
        ok = ok && fwrite(&nn, 8, 1, f) == 1;
        for (const auto& s : camera_names) {
            std::vector<char> c(s.begin(), s.end());
            ok = ok && wr(f, c);
        }
        return fclose(f) == 0 && ok;
    }

    bool load(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        char magic[8];
        bool ok = fread(magic, 8, 1, f) == 1 && std::memcmp(magic, "RTSCENE1", 8) == 0;
        uint32_t sizes[12] = {0};
        ok = ok && fread(sizes, sizeof(sizes), 1, f) == 1;
        ok = ok && sizes[0] == sizeof(rt_bvh_node) && sizes[1] == sizeof(rt_top_prim) &&
             sizes[2] == sizeof(rt_mesh_bvh) && sizes[3] == sizeof(rt_face) && sizes[4] == sizeof(rt_vertex) &&
             sizes[5] == sizeof(rt_mesh) && sizes[6] == sizeof(rt_triangle) && sizes[7] == sizeof(rt_sphere) &&
             sizes[8] == sizeof(rt_plane) && sizes[9] == sizeof(rt_material) &&
             sizes[10] == sizeof(rt_point_light) && sizes[11] == sizeof(rt_camera_desc);
        ok = ok && fread(background_color, 12, 1, f) == 1 && fread(ambient_light, 12, 1, f) == 1;
        ok = ok && fread(&shadow_ray_epsilon, 4, 1, f) == 1 && fread(&max_recursion_depth, 4, 1, f) == 1;
        ok = ok && rd(f, top_nodes) && rd(f, top_prim_idx) && rd(f, top_prims) && rd(f, mesh_bvhs) &&
             rd(f, mesh_nodes) && rd(f, mesh_prim_face) && rd(f, faces) && rd(f, vertex_data) && rd(f, meshes) &&
             rd(f, triangles) && rd(f, spheres) && rd(f, planes) && rd(f, materials) && rd(f, point_lights) &&
             rd(f, area_lights) && rd(f, cameras);
        uint64_t nn = 0;
        ok = ok && fread(&nn, 8, 1, f) == 1 && nn < 4096;
        camera_names.clear();
        for (uint64_t i = 0; ok && i < nn; i++) {
            std::vector<char> c;
            ok = rd(f, c);
            camera_names.emplace_back(c.begin(), c.end());
        }
        fclose(f);
        // arrays that are indexed by one another's counts must agree (n_top_prims is the length of both top-level
        // arrays, include/rt.h); a camera without a name has no output file
        ok = ok && top_prim_idx.size() == top_prims.size() && cameras.size() == camera_names.size();
        if (ok) finalize();
        return ok;
    }
};
