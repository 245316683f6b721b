// ref_shim.cpp — TEST INFRASTRUCTURE (oracle).  Not part of the product path.
//
// Compiles the UNMODIFIED reference (fatih-ozdal/Raytracer) from where it lies
// under /root/reference into oracle/_ref/libref_oracle.so and exposes a small C
// ABI around its own functions so tests can:
//   * parse a scene JSON with the reference parser and build its BVHs
//     (parser::loadFromJson, BuildAllMeshBVHs, BuildTopLevelBVH —
//     raytracer.cpp:22-26),
//   * render with the reference's own pixel loop (raytracer.cpp:40-87, restated
//     here only as the loop header; the body calls ComputeRay / ComputeColor of
//     the reference),
//   * obtain primary-hit identities straight from IntersectTopBVH
//     (raytracer.cpp:527-756) and the plane loop (raytracer.cpp:423-435),
//   * flatten Scene + BVH globals to an RTSCENE1 blob (rt_flatten.hpp).
//
// The reference defines functions and a global in headers (Bvh.h:17-37,76;
// HitRecord.h:15-23), so it can only be one translation unit: raytracer.cpp is
// included here with its main() renamed.  parser.cpp is compiled separately by
// the Makefile.  No reference source is copied into this repository.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>

#define main ref_main_unused
#include "raytracer.cpp"   // found via -I/root/reference
#undef main

#include "rt_flatten.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
Scene g_scene;
bool g_loaded = false;
double g_last_seconds = 0.0;

void reset_globals() {
    topPrims.clear();
    topPrimIdx.clear();
    topBvhNodes.clear();
    rootNodeIdx = 0;
    nodesUsed = 1;
    meshBVHs.clear();
}
}  // namespace

extern "C" {

int ref_load(const char* json_path) {
    try {
        reset_globals();
        parser p;
        g_scene = p.loadFromJson(json_path);
        BuildAllMeshBVHs(g_scene);
        BuildTopLevelBVH(g_scene);
        g_loaded = true;
        return 0;
    } catch (...) {
        g_loaded = false;
        return -1;
    }
}

int ref_num_cameras(void) { return g_loaded ? (int)g_scene.cameras.size() : -1; }

int ref_camera(int i, rt_camera_desc* out) {
    if (!g_loaded || i < 0 || i >= (int)g_scene.cameras.size()) return -1;
    *out = rt_flatten_camera(g_scene.cameras[i]);
    return 0;
}

int ref_save_flat(const char* path) {
    if (!g_loaded) return -1;
    RtSceneStore st;
    if (!rt_flatten(g_scene, st)) return -2;
    return st.save(path) ? 0 : -3;
}

// Per-pixel driver around the reference's own ComputeRay / ComputeColor /
// IntersectTopBVH.  The draw ORDER is the contract of raytracer.cpp:44-69:
// seed = i*width+j, one std::shuffle of [0,N), then per sample five uniform
// draws (jitter x, jitter y, lens u, lens v, time).  std::mt19937, std::shuffle
// and std::uniform_real_distribution are libstdc++'s, as in the reference build.
struct PixelDraws {
    std::mt19937 gen;
    std::uniform_real_distribution<float> uni{0.0f, 1.0f};
    std::vector<int> lens_cell;
    int side;
    PixelDraws(const Camera& cam, int row, int col) : gen(row * cam.image_width + col), side(cam.samples_per_side) {
        lens_cell.resize(cam.num_samples);
        for (int k = 0; k < cam.num_samples; ++k) lens_cell[k] = k;
        std::shuffle(lens_cell.begin(), lens_cell.end(), gen);
    }
    Ray next_ray(const Scene& scene, const Camera& cam, int row, int col, int s) {
        const float jx = (s % side + uni(gen)) / side;
        const float jy = (s / side + uni(gen)) / side;
        const int cell = lens_cell[s];
        const float lu = (cell % side + uni(gen)) / side;
        const float lv = (cell / side + uni(gen)) / side;
        const float tm = uni(gen);
        return ComputeRay(scene, cam, col, row, jx, jy, lu, lv, tm);
    }
};

// Render a window with the reference's functions.  rgb: tight w*h*3.
// threads <= 0: OpenMP default.  Loop shape as raytracer.cpp:40.
int ref_render(int cam_index, int x0, int y0, int w, int h, unsigned char* rgb, int threads) {
    if (!g_loaded || cam_index < 0 || cam_index >= (int)g_scene.cameras.size()) return -1;
    const Scene& scene = g_scene;
    const Camera& cam = scene.cameras[cam_index];
    const int n = cam.num_samples;
    const float inv_n = 1.0f / n;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for collapse(2) schedule(static)
    for (int row = y0; row < y0 + h; row++) {
        for (int col = x0; col < x0 + w; col++) {
            PixelDraws px(cam, row, col);
            Vec3f acc(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < n; ++s) {
                Ray ray = px.next_ray(scene, cam, row, col, s);
                acc = acc + ComputeColor(ray, scene, cam, px.gen, px.uni);
            }
            const Vec3f c = acc * inv_n;
            unsigned char* o = rgb + (static_cast<size_t>(row - y0) * w + (col - x0)) * 3;
            o[0] = (unsigned char)clampF(c.x, 0.0f, 255.0f);
            o[1] = (unsigned char)clampF(c.y, 0.0f, 255.0f);
            o[2] = (unsigned char)clampF(c.z, 0.0f, 255.0f);
        }
    }
    g_last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

double ref_last_render_seconds(void) { return g_last_seconds; }

int ref_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Primary hit of the FIRST sample of each pixel, straight from the reference's
// IntersectTopBVH and IntersectsPlane.  out: 6 int32 per pixel =
//   kind (-1 = miss), index, face.i0, face.i1, face.i2, bits of minT.
int ref_primary_hits(int cam_index, int x0, int y0, int w, int h, int32_t* out) {
    if (!g_loaded || cam_index < 0 || cam_index >= (int)g_scene.cameras.size()) return -1;
    const Scene& scene = g_scene;
    const Camera& cam = scene.cameras[cam_index];
#pragma omp parallel for collapse(2) schedule(static)
    for (int row = y0; row < y0 + h; row++) {
        for (int col = x0; col < x0 + w; col++) {
            PixelDraws px(cam, row, col);
            Ray ray = px.next_ray(scene, cam, row, col, 0);

            float best = FLT_MAX;
            bool any = false;
            PrimKind kind = PrimKind::Mesh;
            int mat = 0, index = 0, mesh_id = 0;
            float bb = 0, gg = 0;
            bool smooth = false;
            Face face{};
            Sphere sph;
            Triangle tri;
            // planes come first and are outside the BVH (FindClosestHit, raytracer.cpp:423-435)
            for (size_t k = 0; k < scene.planes.size(); k++) {
                float t = IntersectsPlane(ray, scene.planes[k].n_unit, scene.planes[k].plane_d, best);
                if (t < best && t != RAY_MISS_VALUE) { best = t; kind = PrimKind::Plane; any = true; index = (int)k; }
            }
            IntersectTopBVH(ray, scene, best, any, kind, mat, index, face, sph, tri, mesh_id, smooth, bb, gg);
            int32_t* o = out + (static_cast<size_t>(row - y0) * w + (col - x0)) * 6;
            std::memset(o, 0, 6 * sizeof(int32_t));
            if (!any) { o[0] = -1; continue; }
            o[0] = (int32_t)kind;
            o[1] = index;
            if (kind == PrimKind::Mesh || kind == PrimKind::Triangle) { o[2] = face.i0; o[3] = face.i1; o[4] = face.i2; }
            std::memcpy(&o[5], &best, 4);
        }
    }
    return 0;
}

}  // extern "C"
