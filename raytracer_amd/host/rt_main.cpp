// rt_main.cpp — the reference's main() with its pixel loop replaced by the C ABI.
//
// Everything before the loop (parser::loadFromJson, BuildAllMeshBVHs,
// BuildTopLevelBVH — raytracer.cpp:22-26) and after it (stbi_write_png,
// raytracer.cpp:89) is the reference's own code, compiled from /root/reference:
// raytracer.cpp is included as a unit with its main() renamed, because the
// reference defines functions and a global in headers (Bvh.h:17-37,76;
// HitRecord.h:15-23).  What is new is ~40 lines: flatten (rt_flatten.hpp),
// rt_upload_scene once per GPU, rt_render_tile_rows_device per GPU per camera,
// copy the rows back.  Built by oracle/Makefile into oracle/_ref/raytracer_hip
// (it contains reference code, so it is not kept in the repository).
//
//   raytracer_hip scene.json [--gpus N] [--backend hip|cpu] [--dump-hits file] [--near-first] [--rebuild-bvh]
// --near-first / --rebuild-bvh select the library's two opt-in non-parity modes (include/rt.h:
// RT_FLAG_NEAR_FIRST, RT_UPLOAD_REBUILD_MESH_BVH); without them the PNG is the reference's, byte for byte.
//
// --backend cpu runs the reference's own loop (raytracer.cpp:40-87 unchanged in
// behaviour) so both paths can be timed in the same process.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define main ref_main_unused
#include "raytracer.cpp"   // -I/root/reference
#undef main

#include <hip/hip_runtime_api.h>

#include "rt.h"
#include "rt_flatten.hpp"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The reference loop, for --backend cpu (same calls in the same order as raytracer.cpp:42-86).
static void render_cpu(const Scene& scene, const Camera& cam, unsigned char* image) {
    const int w = cam.image_width, h = cam.image_height, n = cam.num_samples, side = cam.samples_per_side;
    const float inv_n = 1.0f / n;
#pragma omp parallel for collapse(2) schedule(static)
    for (int row = 0; row < h; row++)
        for (int col = 0; col < w; col++) {
            std::mt19937 gen(row * w + col);
            std::uniform_real_distribution<float> uni(0.0f, 1.0f);
            std::vector<int> cell(n);
            for (int k = 0; k < n; ++k) cell[k] = k;
            std::shuffle(cell.begin(), cell.end(), gen);
            Vec3f acc(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < n; ++s) {
                const float jx = (s % side + uni(gen)) / side, jy = (s / side + uni(gen)) / side;
                const float lu = (cell[s] % side + uni(gen)) / side, lv = (cell[s] / side + uni(gen)) / side;
                const float tm = uni(gen);
                Ray ray = ComputeRay(scene, cam, col, row, jx, jy, lu, lv, tm);
                acc = acc + ComputeColor(ray, scene, cam, gen, uni);
            }
            const Vec3f c = acc * inv_n;
            unsigned char* o = image + (static_cast<size_t>(row) * w + col) * 3;
            o[0] = (unsigned char)clampF(c.x, 0.0f, 255.0f);
            o[1] = (unsigned char)clampF(c.y, 0.0f, 255.0f);
            o[2] = (unsigned char)clampF(c.z, 0.0f, 255.0f);
        }
}

int main(int argc, char* argv[]) {
    if (argc < 2) {
        printf("Usage: %s <scene.json> [--gpus N] [--backend hip|cpu] [--dump-hits file] [--near-first] [--rebuild-bvh]\n", argv[0]);
        return -1;
    }
    int gpus = 1;
    std::string backend = "hip", dump_hits;
    bool near_first = false, rebuild_bvh = false;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--backend") && i + 1 < argc) backend = argv[++i];
        else if (!strcmp(argv[i], "--dump-hits") && i + 1 < argc) dump_hits = argv[++i];
        else if (!strcmp(argv[i], "--near-first")) near_first = true;
        else if (!strcmp(argv[i], "--rebuild-bvh")) rebuild_bvh = true;
    }

    double t0 = now_s();
    parser p;
    Scene scene = p.loadFromJson(argv[1]);
    double t1 = now_s();
    BuildAllMeshBVHs(scene);
    BuildTopLevelBVH(scene);
    double t2 = now_s();

    std::vector<rt_scene*> dev;
    RtSceneStore flat;
    double t_upload = 0;
    if (backend == "hip") {
        if (!rt_flatten(scene, flat)) { fprintf(stderr, "flatten failed\n"); return -1; }
        int n = rt_device_count();
        if (n <= 0) { fprintf(stderr, "%s\n", rt_last_error()); return -1; }
        if (gpus > n) gpus = n;
        double a = now_s();
        dev.resize(gpus, nullptr);
        for (int g = 0; g < gpus; g++)
        {
            if (rt_upload_scene_ex(g, &flat.desc, rebuild_bvh ? RT_UPLOAD_REBUILD_MESH_BVH : 0u, &dev[g]) != RT_OK) { fprintf(stderr, "upload: %s\n", rt_last_error()); return -1; }
            if (near_first && rt_scene_set_option(dev[g], RT_OPT_FLAGS, RT_FLAG_NEAR_FIRST) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return -1; }
        }
        t_upload = now_s() - a;
    }

    for (size_t ci = 0; ci < scene.cameras.size(); ci++) {
        const Camera& camera = scene.cameras[ci];
        const int width = camera.image_width, height = camera.image_height;
        std::vector<unsigned char> image((size_t)width * height * 3);
        double a = now_s();
        unsigned long long rays = 0;
        if (backend == "hip") {
            const rt_camera_desc cam = rt_flatten_camera(camera);
            std::vector<uint32_t> hits(dump_hits.empty() ? 0 : (size_t)width * height * 2);
            std::vector<rt_stats> st(gpus);
            std::vector<std::thread> th;
            std::vector<int> rc(gpus, 0);
            for (int g = 0; g < gpus; g++)
                th.emplace_back([&, g] {
                    // each GPU: its interleaved tile rows into its own device frame, then only those rows come back
                    void *d_fb = nullptr, *d_hits = nullptr;
                    if (hipSetDevice(g) != hipSuccess || hipMalloc(&d_fb, image.size()) != hipSuccess) { rc[g] = -1; return; }
                    if (!hits.empty() && hipMalloc(&d_hits, hits.size() * 4) != hipSuccess) { rc[g] = -1; return; }
                    rc[g] = rt_render_tile_rows_device(dev[g], &cam, 0, 0, width, height, gpus, g, d_fb, (size_t)width * 3, d_hits, &st[g], nullptr);
                    for (int ty = g; rc[g] == RT_OK && ty * 8 < height; ty += gpus) {
                        const int y = ty * 8, rows = std::min(8, height - y);
                        if (hipMemcpy(&image[(size_t)y * width * 3], (char*)d_fb + (size_t)y * width * 3, (size_t)rows * width * 3, hipMemcpyDeviceToHost) != hipSuccess) rc[g] = -1;
                        if (!hits.empty() && hipMemcpy(&hits[(size_t)y * width * 2], (char*)d_hits + (size_t)y * width * 8, (size_t)rows * width * 8, hipMemcpyDeviceToHost) != hipSuccess) rc[g] = -1;
                    }
                    (void)hipFree(d_fb);
                    if (d_hits) (void)hipFree(d_hits);
                });
            for (auto& t : th) t.join();
            for (int g = 0; g < gpus; g++) {
                if (rc[g] != RT_OK) { fprintf(stderr, "render on GPU %d: %s\n", g, rt_last_error()); return -1; }
                rays += st[g].rays_primary + st[g].rays_secondary + st[g].rays_shadow;
            }
            if (!dump_hits.empty()) {
                FILE* f = fopen(dump_hits.c_str(), "wb");
                if (f) { fwrite(hits.data(), 4, hits.size(), f); fclose(f); }
            }
        } else {
            render_cpu(scene, camera, image.data());
        }
        double sec = now_s() - a;
        stbi_write_png(camera.image_name.c_str(), width, height, 3, image.data(), width * 3);   // raytracer.cpp:89
        printf("{\"camera\": \"%s\", \"backend\": \"%s\", \"gpus\": %d, \"width\": %d, \"height\": %d, \"spp\": %d, \"render_s\": %.6f, "
               "\"rays\": %llu, \"mrays_per_s\": %.3f, \"parse_s\": %.3f, \"bvh_s\": %.3f, \"upload_s\": %.3f}\n",
               camera.image_name.c_str(), backend.c_str(), backend == "hip" ? gpus : 0, width, height, camera.num_samples, sec, rays,
               rays ? rays / sec / 1e6 : 0.0, t1 - t0, t2 - t1, t_upload);
    }
    for (rt_scene* s : dev) rt_free_scene(s);
    return 0;
}
