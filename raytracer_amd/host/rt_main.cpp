// rt_main.cpp — the reference's main() with its pixel loop replaced by the C ABI, grown into the frame
// driver of SURVEY §8f-4.
//
// Everything before the loop (parser::loadFromJson, BuildAllMeshBVHs, BuildTopLevelBVH — raytracer.cpp:22-26)
// and after it (stbi_write_png, raytracer.cpp:89) is the reference's own code, compiled from /root/reference:
// raytracer.cpp is included as a unit with its main() renamed, because the reference defines functions and a
// global in headers (Bvh.h:17-37,76; HitRecord.h:15-23).  Built by oracle/Makefile into
// oracle/_ref/raytracer_hip (it contains reference code, so the binary is not kept in the repository).
//
//   raytracer_hip scene.json [more.json ...] [--frames PREFIX START END] [--gpus N] [--backend hip|cpu]
//                 [--tile WxH] [--bench K] [--serial] [--dump-hits file] [--near-first] [--rebuild-bvh]
//
// One JSON: the reference's behaviour — one PNG per camera, named by the camera (raytracer.cpp:89).
// Several JSONs, or --frames PREFIX START END (the files PREFIX%03d.json that exist, exactly the loop of
// render_and_make_gif.sh:102-115, which starts one ./raytracer PROCESS per frame): the frames of an animation
// go through a three-stage pipeline on host threads —
//     parse + BuildAllMeshBVHs + BuildTopLevelBVH + flatten of frame k+1      (reference code, its globals)
//  || upload-or-update + render of frame k on the GPU(s)                      (C ABI)
//  || stbi_write_png of frame k-1                                             (reference's writer)
// — and the device scene stays RESIDENT: when the next frame's mesh arrays are the ones already uploaded
// (sizes, then the 64-bit hash: RT_UPDATE_VERIFY_MESHES) only rt_update_scene runs (transforms, lights,
// materials, top-level BVH; the camera is a render argument anyway); otherwise the scene is uploaded afresh.
// --serial runs the three stages one after the other (the baseline the pipeline is measured against).
//
// --gpus N: every GPU holds a replica of the scene and renders the interleaved 8-row bands g mod N of the frame
// into its own device frame (rt_render_tile_rows_device); each GPU then sends its bands with ONE strided
// device-to-device copy over its xGMI link into GPU 0's frame (peer access enabled), and GPU 0 sends the
// finished frame to pinned host memory in one copy — the message shape is the reference's image,
// width*height*3 bytes, rows top first (raytracer.cpp:34,82-85).
// --tile WxH: the frame is rendered as WxH rectangles (one call each) instead of one call; same bytes.
// --bench K: each camera is rendered K more times; the JSON line reports the mean of those.
// --near-first / --rebuild-bvh select the library's two opt-in non-parity modes (include/rt.h); without them
// the PNG is the reference's, byte for byte.  --backend cpu runs the reference's own loop (raytracer.cpp:40-87
// unchanged in behaviour) so both paths can be timed in the same process.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
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

// ---------------------------------------------------------------------------------------------- pipeline plumbing
template <class T>
struct Channel {          // bounded hand-off between two stages; close() ends the consumer's loop
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap;
    bool closed = false;
    explicit Channel(size_t c) : cap(c) {}
    void push(T v) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.size() < cap; });
        q.push_back(std::move(v));
        cv.notify_all();
    }
    bool pop(T& out) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        out = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m);
        closed = true;
        cv.notify_all();
    }
};

struct ParsedFrame {      // what stage 1 hands to stage 2
    std::string json;
    Scene scene;          // kept for --backend cpu (the reference's loop reads it and the BVH globals)
    RtSceneStore flat;
    double parse_s = 0, bvh_s = 0;
    bool ok = false;
};
struct PinnedImage {      // what stage 2 hands to stage 3
    unsigned char* px = nullptr;
    size_t bytes = 0;
    bool pinned = false;
    ~PinnedImage() {
        if (!px) return;
        if (pinned) (void)hipHostFree(px); else free(px);
    }
    bool ensure(size_t n, bool want_pinned) {
        if (n <= bytes) return true;
        if (px) { if (pinned) (void)hipHostFree(px); else free(px); px = nullptr; bytes = 0; }
        if (want_pinned && hipHostMalloc((void**)&px, n) == hipSuccess) { pinned = true; bytes = n; return true; }
        px = (unsigned char*)malloc(n);
        pinned = false;
        bytes = px ? n : 0;
        return px != nullptr;
    }
};
struct RenderedImage {
    std::string name;
    int width = 0, height = 0;
    std::unique_ptr<PinnedImage> img;
};

static void reset_reference_globals() {   // the reference is a one-shot program: its builders append to these
    topPrims.clear();
    topPrimIdx.clear();
    topBvhNodes.clear();
    rootNodeIdx = 0;
    nodesUsed = 1;
    meshBVHs.clear();
}

static void parse_frame(const std::string& json, bool flatten, ParsedFrame& f) {
    f.json = json;
    const double t0 = now_s();
    reset_reference_globals();
    parser p;
    f.scene = p.loadFromJson(json);
    const double t1 = now_s();
    BuildAllMeshBVHs(f.scene);
    BuildTopLevelBVH(f.scene);
    f.ok = !flatten || rt_flatten(f.scene, f.flat);
    f.parse_s = t1 - t0;
    f.bvh_s = now_s() - t1;
}

// ---------------------------------------------------------------------------------------------- the GPUs of one node
struct Gpu {
    int device = 0;           // HIP device (RT_MAIN_ONE_DEVICE=1 maps every logical GPU to device 0: rehearsal on a 1-GPU box)
    rt_scene* scene = nullptr;
    void* d_fb = nullptr;     // full-size device frame of this GPU (only its bands are written)
    void* d_hits = nullptr;
    size_t fb_bytes = 0, hits_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t sent = nullptr;
    std::string err;          // rt_last_error() is thread-local: captured where the call was made
    rt_stats st{};
};

struct Options {
    int gpus = 1, bench = 0, tile_w = 0, tile_h = 0;
    bool near_first = false, rebuild_bvh = false, rebuild_on_device = false, rebuild_lbvh = false, serial = false;
    std::string backend = "hip", dump_hits;
};

static bool gpu_buffers(Gpu& G, size_t fb_bytes, size_t hits_bytes) {
    if (hipSetDevice(G.device) != hipSuccess) { G.err = "hipSetDevice failed"; return false; }
    if (fb_bytes > G.fb_bytes) {
        if (G.d_fb) (void)hipFree(G.d_fb);
        G.d_fb = nullptr; G.fb_bytes = 0;
        if (hipMalloc(&G.d_fb, fb_bytes) != hipSuccess) { G.err = "hipMalloc of the device frame failed"; return false; }
        G.fb_bytes = fb_bytes;
    }
    if (hits_bytes > G.hits_bytes) {
        if (G.d_hits) (void)hipFree(G.d_hits);
        G.d_hits = nullptr; G.hits_bytes = 0;
        if (hipMalloc(&G.d_hits, hits_bytes) != hipSuccess) { G.err = "hipMalloc of the hit buffer failed"; return false; }
        G.hits_bytes = hits_bytes;
    }
    return true;
}

// Bands of GPU g (8-row bands ty = g, g + n, ...) of a `rows`-row image, as one strided 2-D copy of the full bands
// plus, for the owner of a shorter last band, one more copy.  bytes_per_row = bytes of one image row.
static hipError_t copy_bands(void* dst, const void* src, size_t bytes_per_row, int rows, int n, int g, hipMemcpyKind kind, hipStream_t st) {
    const int bands_all = (rows + 7) / 8;
    if (g >= bands_all) return hipSuccess;
    const int mine = (bands_all - g + n - 1) / n;                   // bands of this GPU
    const bool last_is_mine = (bands_all - 1) % n == g;
    const int last_rows = rows - (bands_all - 1) * 8;               // 1..8
    const int full = (last_is_mine && last_rows < 8) ? mine - 1 : mine;
    const size_t band = 8 * bytes_per_row, pitch = (size_t)n * band, off0 = (size_t)g * band;
    hipError_t e = hipSuccess;
    if (full > 0) e = hipMemcpy2DAsync((char*)dst + off0, pitch, (const char*)src + off0, pitch, band, (size_t)full, kind, st);
    if (e == hipSuccess && full < mine) {
        const size_t off = (size_t)(bands_all - 1) * band;
        e = hipMemcpyAsync((char*)dst + off, (const char*)src + off, (size_t)last_rows * bytes_per_row, kind, st);
    }
    return e;
}

// One camera on all GPUs: render, bands to GPU 0 over peer copies, the frame to `image` (pinned).  Returns rays traced.
static bool render_frame_hip(std::vector<Gpu>& gpus, const rt_camera_desc& cam, const Options& opt, unsigned char* image, uint32_t* hits,
                             unsigned long long& rays, double& render_s) {
    const int n = (int)gpus.size(), W = cam.image_width, H = cam.image_height;
    const size_t row = (size_t)W * 3, fb_bytes = row * H, hits_bytes = hits ? (size_t)W * H * 8 : 0;
    for (Gpu& G : gpus)
        if (!gpu_buffers(G, fb_bytes, hits_bytes)) return false;
    // tiles: whole frame, or WxH rectangles whose height keeps the bands of a rectangle on the frame's band grid
    int tw = opt.tile_w > 0 ? opt.tile_w : W, th = opt.tile_h > 0 ? opt.tile_h : H;
    if (th < H) th = ((th + 8 * n - 1) / (8 * n)) * (8 * n);
    const double a = now_s();
    std::vector<std::thread> th_gpu;
    std::vector<int> ok(n, 1);
    for (int g = 0; g < n; g++)
        th_gpu.emplace_back([&, g] {
            Gpu& G = gpus[g];
            if (hipSetDevice(G.device) != hipSuccess) { G.err = "hipSetDevice failed"; ok[g] = 0; return; }
            std::memset(&G.st, 0, sizeof(G.st));
            for (int y0 = 0; y0 < H && ok[g]; y0 += th)
                for (int x0 = 0; x0 < W && ok[g]; x0 += tw) {
                    rt_stats st;
                    const int w = std::min(tw, W - x0), h = std::min(th, H - y0);
                    const int rc = rt_render_tile_rows_device(G.scene, &cam, x0, y0, w, h, n, g, (char*)G.d_fb + (size_t)y0 * row + (size_t)x0 * 3, row,
                                                              hits ? (char*)G.d_hits + ((size_t)y0 * W + x0) * 8 : nullptr, &st, G.stream);
                    if (rc != RT_OK) { G.err = rt_last_error(); ok[g] = 0; break; }
                    G.st.rays_primary += st.rays_primary; G.st.rays_secondary += st.rays_secondary; G.st.rays_shadow += st.rays_shadow;
                }
            if (!ok[g]) return;
            // hit ids are a debugging output: straight to the host, band by band in one strided copy
            if (hits && copy_bands(hits, G.d_hits, (size_t)W * 8, H, n, g, hipMemcpyDeviceToHost, G.stream) != hipSuccess) { G.err = "copy of the hit ids failed"; ok[g] = 0; return; }
        });
    for (auto& t : th_gpu) t.join();
    for (int g = 0; g < n; g++)
        if (!ok[g]) return false;
    // frame assembly (rt.h): every GPU's bands into GPU 0's frame — one strided device-to-device copy per GPU over its
    // direct xGMI link, behind its render on its stream — and the whole frame to the host in one copy
    Gpu& G0 = gpus[0];
    std::vector<rt_scene*> scenes(n);
    std::vector<void*> fbs(n), streams(n);
    for (int g = 0; g < n; g++) { scenes[g] = gpus[g].scene; fbs[g] = gpus[g].d_fb; streams[g] = (void*)gpus[g].stream; }
    if (rt_gather_rows(scenes.data(), n, fbs.data(), row, W, H, streams.data(), image) != RT_OK) { G0.err = rt_last_error(); return false; }
    for (int g = 1; g < n && hits; g++)
        if (hipStreamSynchronize(gpus[g].stream) != hipSuccess) { gpus[g].err = "stream sync failed"; return false; }
    render_s = now_s() - a;
    rays = 0;
    for (Gpu& G : gpus) rays += G.st.rays_primary + G.st.rays_secondary + G.st.rays_shadow;
    return true;
}

// upload-or-update of one frame's scene on every GPU; returns "upload", "update" or "" on failure
static const char* place_scene(std::vector<Gpu>& gpus, const RtSceneStore& flat, const Options& opt, bool& have_scene) {
    const char* how = "update";
    for (Gpu& G : gpus) {
        if (hipSetDevice(G.device) != hipSuccess) { G.err = "hipSetDevice failed"; return ""; }
        int rc = RT_ERR_INVALID;
        if (have_scene && G.scene) rc = rt_update_scene(G.scene, &flat.desc, RT_UPDATE_VERIFY_MESHES);
        if (rc != RT_OK) {                       // other meshes than the resident ones (or the first frame): a fresh upload
            how = "upload";
            if (G.scene) { rt_free_scene(G.scene); G.scene = nullptr; }
            rc = rt_upload_scene_ex(G.device, &flat.desc, opt.rebuild_bvh ? (RT_UPLOAD_REBUILD_MESH_BVH | (opt.rebuild_on_device ? RT_UPLOAD_REBUILD_ON_DEVICE : 0u) | (opt.rebuild_lbvh ? RT_UPLOAD_REBUILD_LBVH : 0u)) : 0u, &G.scene);
            if (rc == RT_OK && opt.near_first) rc = rt_scene_set_option(G.scene, RT_OPT_FLAGS, RT_FLAG_NEAR_FIRST);
            if (rc != RT_OK) { G.err = rt_last_error(); return ""; }
        }
    }
    have_scene = true;
    return how;
}

int main(int argc, char* argv[]) {
    if (argc < 2) {
        printf("Usage: %s <scene.json> [more.json ...] [--frames PREFIX START END] [--gpus N] [--backend hip|cpu] [--tile WxH] [--bench K] "
               "[--serial] [--dump-hits file] [--near-first] [--rebuild-bvh[=device|=lbvh]]\n", argv[0]);
        return -1;
    }
    Options opt;
    std::vector<std::string> jsons;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) opt.gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--backend") && i + 1 < argc) opt.backend = argv[++i];
        else if (!strcmp(argv[i], "--dump-hits") && i + 1 < argc) opt.dump_hits = argv[++i];
        else if (!strcmp(argv[i], "--near-first")) opt.near_first = true;
        else if (!strcmp(argv[i], "--rebuild-bvh")) opt.rebuild_bvh = true;
        else if (!strcmp(argv[i], "--rebuild-bvh=device")) opt.rebuild_bvh = opt.rebuild_on_device = true;   // rt_dsah.h: the same SAH trees built on the GPU
        else if (!strcmp(argv[i], "--rebuild-bvh=lbvh")) opt.rebuild_bvh = opt.rebuild_on_device = opt.rebuild_lbvh = true;   // rt_lbvh.h: linear BVHs built on the GPU
        else if (!strcmp(argv[i], "--serial")) opt.serial = true;
        else if (!strcmp(argv[i], "--bench") && i + 1 < argc) opt.bench = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tile") && i + 1 < argc) {
            if (sscanf(argv[++i], "%dx%d", &opt.tile_w, &opt.tile_h) != 2 || opt.tile_w < 1 || opt.tile_h < 1) { fprintf(stderr, "--tile WxH\n"); return -1; }
        } else if (!strcmp(argv[i], "--frames") && i + 3 < argc) {
            // render_and_make_gif.sh:102-104: printf "${SCENE_PATTERN}%03d.json" for the frames that exist
            const std::string prefix = argv[i + 1];
            const int first = atoi(argv[i + 2]), last = atoi(argv[i + 3]);
            i += 3;
            for (int k = first; k <= last; k++) {
                char num[16];
                snprintf(num, sizeof(num), "%03d", k);
                const std::string path = prefix + num + ".json";
                if (FILE* f = fopen(path.c_str(), "rb")) { fclose(f); jsons.push_back(path); }
            }
        } else if (argv[i][0] == '-' && argv[i][1] == '-') { fprintf(stderr, "unknown option %s\n", argv[i]); return -1; }
        else jsons.push_back(argv[i]);
    }
    if (jsons.empty()) { fprintf(stderr, "no scene files\n"); return -1; }
    const bool hip = opt.backend == "hip";

    std::vector<Gpu> gpus;
    if (hip) {
        int n = rt_device_count();
        if (n <= 0) { fprintf(stderr, "%s\n", rt_last_error()); return -1; }
        const bool one_device = getenv("RT_MAIN_ONE_DEVICE") != nullptr;
        if (opt.gpus < 1) opt.gpus = 1;
        if (!one_device && opt.gpus > n) opt.gpus = n;
        gpus.resize(opt.gpus);
        for (int g = 0; g < opt.gpus; g++) {
            Gpu& G = gpus[g];
            G.device = one_device ? 0 : g;
            if (hipSetDevice(G.device) != hipSuccess || hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&G.sent, hipEventDisableTiming) != hipSuccess) { fprintf(stderr, "GPU %d: stream set-up failed\n", g); return -1; }
            if (g != 0 && G.device != gpus[0].device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, G.device, gpus[0].device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(gpus[0].device, 0);
                (void)hipGetLastError();          // (already enabled is fine; without peer access the runtime stages the copy)
            }
        }
    }

    // ---- stage 1: parse + build + flatten (the reference's code works on its globals: one frame at a time)
    Channel<std::unique_ptr<ParsedFrame>> parsed(opt.serial ? 1 : 2);
    Channel<RenderedImage> rendered(opt.serial ? 1 : 3);
    std::thread t_parse, t_png;
    const double t_begin = now_s();
    // --backend cpu reads the reference's globals while rendering, so the next parse must not start before that
    // frame is done: the cpu backend (and --serial) parse on the render thread instead
    const bool threaded = hip && !opt.serial && jsons.size() > 1;
    if (threaded)
        t_parse = std::thread([&] {
            for (const std::string& j : jsons) {
                std::unique_ptr<ParsedFrame> f(new ParsedFrame());
                parse_frame(j, true, *f);
                parsed.push(std::move(f));
            }
            parsed.close();
        });
    // ---- stage 3: PNG encode + write (raytracer.cpp:89)
    double png_s = 0;
    auto write_png = [&](RenderedImage& r) {
        const double a = now_s();
        stbi_write_png(r.name.c_str(), r.width, r.height, 3, r.img->px, r.width * 3);
        png_s += now_s() - a;
    };
    if (threaded)
        t_png = std::thread([&] {
            RenderedImage r;
            while (rendered.pop(r)) write_png(r);
        });

    // ---- stage 2 (this thread): place the scene, render every camera
    bool have_scene = false, failed = false;
    size_t frames_done = 0, images_done = 0;
    double render_total = 0, place_total = 0;
    for (size_t fi = 0; fi < jsons.size() && !failed; fi++) {
        std::unique_ptr<ParsedFrame> f;
        if (threaded) {
            if (!parsed.pop(f)) break;
        } else {
            f.reset(new ParsedFrame());
            parse_frame(jsons[fi], hip, *f);
        }
        if (!f->ok) { fprintf(stderr, "%s: flatten failed\n", f->json.c_str()); failed = true; break; }
        const char* how = "-";
        double place_s = 0;
        if (hip) {
            const double a = now_s();
            how = place_scene(gpus, f->flat, opt, have_scene);
            place_s = now_s() - a;
            place_total += place_s;
            if (!*how) {
                for (size_t g = 0; g < gpus.size(); g++)
                    if (!gpus[g].err.empty()) fprintf(stderr, "scene on GPU %zu: %s\n", g, gpus[g].err.c_str());
                failed = true;
                break;
            }
        }
        for (size_t ci = 0; ci < f->scene.cameras.size() && !failed; ci++) {
            const Camera& camera = f->scene.cameras[ci];
            const int width = camera.image_width, height = camera.image_height;
            RenderedImage out;
            out.name = camera.image_name;
            out.width = width; out.height = height;
            out.img.reset(new PinnedImage());
            if (!out.img->ensure((size_t)width * height * 3, hip)) { fprintf(stderr, "host frame allocation failed\n"); failed = true; break; }
            unsigned long long rays = 0;
            double sec = 0, bench_s = 0;
            if (hip) {
                const rt_camera_desc cam = rt_flatten_camera(camera);
                const bool want_hits = !opt.dump_hits.empty() && fi == 0 && ci == 0;
                std::vector<uint32_t> hits(want_hits ? (size_t)width * height * 2 : 0);
                bool ok = render_frame_hip(gpus, cam, opt, out.img->px, want_hits ? hits.data() : nullptr, rays, sec);
                for (int k = 0; ok && k < opt.bench; k++) {
                    double s = 0;
                    ok = render_frame_hip(gpus, cam, opt, out.img->px, nullptr, rays, s);
                    bench_s += s;
                }
                if (!ok) {
                    for (size_t g = 0; g < gpus.size(); g++)
                        if (!gpus[g].err.empty()) fprintf(stderr, "render on GPU %zu: %s\n", g, gpus[g].err.c_str());
                    failed = true;
                    break;
                }
                if (want_hits) {
                    FILE* fp = fopen(opt.dump_hits.c_str(), "wb");
                    if (fp) { fwrite(hits.data(), 4, hits.size(), fp); fclose(fp); }
                }
            } else {
                const double a = now_s();
                render_cpu(f->scene, camera, out.img->px);
                sec = now_s() - a;
            }
            render_total += sec;
            if (opt.bench > 0) sec = bench_s / opt.bench;
            printf("{\"scene\": \"%s\", \"camera\": \"%s\", \"backend\": \"%s\", \"gpus\": %d, \"width\": %d, \"height\": %d, \"spp\": %d, \"render_s\": %.6f, "
                   "\"rays\": %llu, \"mrays_per_s\": %.3f, \"parse_s\": %.3f, \"bvh_s\": %.3f, \"scene_on_device\": \"%s\", \"place_s\": %.4f}\n",
                   f->json.c_str(), camera.image_name.c_str(), opt.backend.c_str(), hip ? opt.gpus : 0, width, height, camera.num_samples, sec, rays,
                   rays ? rays / sec / 1e6 : 0.0, f->parse_s, f->bvh_s, how, place_s);
            fflush(stdout);
            images_done++;
            if (threaded) rendered.push(std::move(out)); else write_png(out);
        }
        frames_done++;
    }
    rendered.close();
    if (threaded) {
        if (failed) { std::unique_ptr<ParsedFrame> f; while (parsed.pop(f)) {} }   // let the parser thread run out
        t_parse.join();
        t_png.join();
    }
    const double wall = now_s() - t_begin;
    if (jsons.size() > 1)
        printf("{\"frames\": %zu, \"images\": %zu, \"wall_s\": %.4f, \"frames_per_s\": %.3f, \"pipeline\": %s, \"render_s_total\": %.4f, "
               "\"scene_placement_s_total\": %.4f, \"png_s_total\": %.4f}\n",
               frames_done, images_done, wall, frames_done / wall, threaded ? "true" : "false", render_total, place_total, png_s);
    for (Gpu& G : gpus) {
        (void)hipSetDevice(G.device);
        if (G.scene) rt_free_scene(G.scene);
        if (G.d_fb) (void)hipFree(G.d_fb);
        if (G.d_hits) (void)hipFree(G.d_hits);
        if (G.sent) (void)hipEventDestroy(G.sent);
        if (G.stream) (void)hipStreamDestroy(G.stream);
    }
    return failed ? -1 : 0;
}
