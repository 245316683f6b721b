// rt_ordered.h — the pixel loop for scenes whose SHADING draws random numbers:
// glossy reflection / refraction (PerturbReflection, raytracer.cpp:1438-1457) and
// area lights (raytracer.cpp:1379-1433).  SURVEY.md §8f-1.
//
// There the per-pixel generator (raytracer.cpp:44) is consumed in the order of the
// recursion: 5 draws per sample (:59-69), then inside ApplyShading 2 draws before a
// rough reflection, the whole reflect subtree, 2 draws before a rough refraction, the
// whole refract subtree, and 2 draws per area light (:1382-1383).  How many draws a
// subtree takes depends on what its rays hit, so the rays of one pixel cannot be
// generated generation by generation as rt_kernels.h does; they are strictly
// sequential per pixel.  Pixels stay independent.
//
// So here ONE LANE OWNS ONE PIXEL and walks ComputeColor / ApplyShading depth-first
// with an explicit frame stack in global memory (the reference recurses,
// raytracer.cpp:382-401, 1248-1436), with its mt19937 state in global scratch.
// Whenever a lane needs FindClosestHit or InShadow it posts the ray in registers and
// waits; the WAVE then runs the same persistent traversal loop as the other kernels
// (rt_traverse.h: trace_persistent) over its own 64 lanes as items — so idle lanes
// still take over subtrees of the busy ones — and every lane resumes with its result.
// The colour arithmetic is the same device code (shade_ctx, diffuse_specular, ...)
// in the reference's expression order.
#pragma once
#include "rt_kernels.h"

struct DAreaLight { float px, py, pz, nx, ny, nz, size, rx, ry, rz; };

// frame fields (one float/uint each, stored [frame][field][thread])
enum {
    OF_OX, OF_OY, OF_OZ, OF_DX, OF_DY, OF_DZ,     // the frame's ray
    OF_THIT, OF_NX, OF_NY, OF_NZ, OF_MAT,         // its hit: t, HitRecord::normal, material id
    OF_STAGE, OF_LIGHT,                           // where ApplyShading stands; first light of the batch of tests in flight
    OF_CR, OF_CG, OF_CB,                          // `color` so far
    OF_RR, OF_RG, OF_RB,                          // reflectColor (dielectric, raytracer.cpp:1335)
    OF_COUNT
};
enum { OS_TRACE = 0, OS_REFLECT_WAIT, OS_REFRACT_WAIT, OS_LIGHTS };
enum { OREQ_NONE = 0, OREQ_CLOSEST = 1, OREQ_ANYHIT = 2 };

// ---- pixel state and request slots.  Everything a pixel needs between two of its rays lives in global memory, one
// 4-byte field per slot, stored [field][slot] so that the 64 lanes of a wave read and write neighbouring words: the
// pixel's place in the loop below (PS_*), its frame stack (OF_*), its generator.  The REQUESTS a pixel has posted and
// the ANSWERS it gets back are fields too, RT_ORDERED_REQS of them per lane (request r of lane l = item r * 64 + l of
// the wave), so any lane of the wave can trace any request: the traversal loop sees the wave's 64 * REQS requests as
// one list of items with a cursor and refills idle lanes from it.
//
// Why several requests per pixel: the rays of a pixel are sequential only where a result decides what is drawn or
// traced next.  The light tests of one hit are not: InShadow towards each point light and the closest hit towards each
// area-light sample (raytracer.cpp:1371-1433) only add to `color`, and the two draws per area light follow each other
// with nothing drawn in between.  So a hit posts all its light tests at once (REQS at a time), the draws made in the
// reference's order when they are posted, and the contributions are added in the reference's order when the answers
// are back: the chain of rounds per hit is 1 + ceil(lights / REQS) instead of 1 + lights.
#ifndef RT_ORDERED_REQS
#define RT_ORDERED_REQS 2
#endif
enum {
    // per pixel (request slot 0 of the lane holds them)
    PS_IDX,                                     // pixel index in the chunk; 0xFFFFFFFF: the lane has no pixel
    PS_FLAGS,                                   // 1: ret_pending, 2: start_sample
    PS_S, PS_DEPTH,
    PS_ACC_R, PS_ACC_G, PS_ACC_B,
    PS_RET_R, PS_RET_G, PS_RET_B,
    PS_HIT_W0, PS_HIT_TRI,
    PS_GEN_IDX, PS_GEN_LO, PS_GEN_HI, PS_TIME,
    // per request slot
    PS_REQ, PS_OX, PS_OY, PS_OZ, PS_DX, PS_DY, PS_DZ, PS_DIST,          // the request: OREQ_*, ray, InShadow's distance
    PS_HAS, PS_T, PS_KIND, PS_TRI, PS_BETA, PS_GAMMA,                   // the answer
    PS_COUNT
};
#define RT_SLOT_EMPTY 0xFFFFFFFFu

// The request a slot has posted, and the answer it gets back (registers while a lane works on the slot).
struct LaneTrace {
    int req;                     // OREQ_*
    V3 o, d;
    float time, dist;            // dist: InShadow's minT = distance to the light
    bool has;                    // closest: hit found / any-hit: in shadow
    float t;
    uint32_t kind_index, tri;
    float beta, gamma;
};

// The wave's requests as items of the trace loop (trace_persistent MODE 2): item i = slot i of this wave.
struct SlotMixedIO {
    const DScene& S;
    uint32_t* ps;                // this wave's slot 0 of field 0
    size_t n_slots;              // words between fields
    __device__ __forceinline__ uint32_t& f(uint32_t field, uint32_t i) const { return ps[(size_t)field * n_slots + i]; }
    __device__ __forceinline__ float ff(uint32_t field, uint32_t i) const { return __uint_as_float(ps[(size_t)field * n_slots + i]); }
    __device__ __forceinline__ uint32_t first_blocks() const { return 1u; }      // the cursor hands out slot 0 of every lane first, then the other slots in order
    __device__ __forceinline__ uint32_t first_count() const { return 64u; }
    __device__ __forceinline__ int load(uint32_t i, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h, bool& any) const {
        const uint32_t req = f(PS_REQ, i);
        if (req == OREQ_NONE) return 0;
        o = mk(ff(PS_OX, i), ff(PS_OY, i), ff(PS_OZ, i));
        d = mk(ff(PS_DX, i), ff(PS_DY, i), ff(PS_DZ, i));
        time = ff(PS_TIME, i & 63u);            // the pixel's (raytracer.cpp:69), kept with its lane
        has = false;
        h.kind_index = 0xFFFFFFFFu; h.tri = 0; h.beta = 0.f; h.gamma = 0.f;
        if (req == OREQ_CLOSEST) {
            any = false;
            minT = RT_FLT_MAX;
            for (uint32_t k = 0; k < S.n_planes; k++) {   // raytracer.cpp:423-435
                float t;
                if (plane_test(o, d, S.planes[k], minT, t)) { minT = t; has = true; h.kind_index = (RT_KIND_PLANE << 28) | k; }
            }
            return 2;
        }
        any = true;
        minT = ff(PS_DIST, i);
        for (uint32_t k = 0; k < S.n_planes; k++) {       // raytracer.cpp:1022-1028
            float t;
            if (plane_test(o, d, S.planes[k], minT, t)) { f(PS_HAS, i) = 1u; f(PS_REQ, i) = OREQ_NONE; return 1; }
        }
        return 2;
    }
    __device__ __forceinline__ void store(uint32_t i, bool verdict, const HitOut& h, float minT) const {
        if (f(PS_REQ, i) == OREQ_ANYHIT) {
            f(PS_HAS, i) = verdict ? 1u : 0u;
        } else {
            f(PS_HAS, i) = verdict ? 1u : 0u;
            f(PS_T, i) = __float_as_uint(minT);
            f(PS_KIND, i) = verdict ? h.kind_index : 0xFFFFFFFFu;
            f(PS_TRI, i) = h.tri;
            f(PS_BETA, i) = __float_as_uint(h.beta);
            f(PS_GAMMA, i) = __float_as_uint(h.gamma);
        }
        f(PS_REQ, i) = OREQ_NONE;
    }
};

struct OrderedArgs {
    DCamera cam;
    ChunkGeom g;
    const DAreaLight* area_lights;
    uint32_t n_area_lights;
    uint32_t* mt_scratch;        // [624 + N][threads]
    float* frames;               // [max_depth + 1][OF_COUNT][threads]
    uint32_t* pstate;            // [PS_COUNT][threads * RT_ORDERED_REQS]
    float inv_n;
    uint8_t* rgb;
    size_t row_stride;
    uint32_t* hit_ids;
    const DTriShade* tri_shade;
    const uint32_t* tile_order;  // NULL, or the order in which the cursor hands out the chunk's 8x8 tiles (costliest first, k_tile_cost)
};

// What a tile will cost, from the closest hit of one ray per pixel (the records of a k_raygen_single / k_trace_closest
// pass over the chunk): pixels that open a refraction tree weigh most.  The launch is as long as its costliest pixels'
// chains of sequential rays, so those pixels start first; the order changes no pixel's rays.
__global__ void __launch_bounds__(256) k_tile_cost(DScene S, Pool P, uint32_t n_idx, uint32_t* cost) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w = 0;
    if (idx < n_idx) {
        const uint4 h0 = P.hit0[idx];
        w = 1;
        if (h0.y != 0xFFFFFFFFu) {
            const uint32_t kind = h0.y >> 28, index = h0.y & 0x0FFFFFFFu;
            int mid = 1;
            if (kind == RT_KIND_MESH) mid = S.objects[S.mesh_obj_base + index].material_id;
            else if (kind == RT_KIND_TRIANGLE) mid = S.objects[S.tri_obj_base + index].material_id;
            else if (kind == RT_KIND_SPHERE) mid = S.objects[S.sphere_obj_base + index].material_id;
            else mid = S.planes[index].material_id;
            const int type = S.materials[mid - 1].type;
            w = type == RT_MAT_DIELECTRIC ? 24u : ((type == RT_MAT_MIRROR || type == RT_MAT_CONDUCTOR) ? 6u : 3u);
        }
    }
    for (int off = 32; off; off >>= 1) w += (uint32_t)__shfl_xor((int)w, off, 64);
    if ((threadIdx.x & 63u) == 0 && idx < n_idx) cost[idx >> 6] = w;
}

#ifndef RT_ORDERED_MIN_BLOCKS
#define RT_ORDERED_MIN_BLOCKS 2  // workgroups per CU the register budget is set for: 256 VGPRs, 63 spilled (3: 168 VGPRs, 339 spilled, 10 % slower; 4: 22 % slower; 1: 26 % slower)
#endif
template <bool NEAR>
__global__ void __launch_bounds__(RT_ORDERED_BLOCK, RT_ORDERED_MIN_BLOCKS) k_pixel_ordered(DScene S, OrderedArgs A, Control* ctl, uint2* spill, uint4* cand) {
    __shared__ TraceLds<RT_ORDERED_BLOCK> s_lds;
    uint32_t* s_cursor = s_lds.cursor;
    WaveWork ww;
    TraceStack st = make_stack(s_lds, spill, cand, ww, S.stack_cap);
#ifdef RT_TRACE_STATS
    TraceStatsAcc acc = {};
    st.acc = &acc;
#endif
    constexpr uint32_t K = RT_ORDERED_REQS;
    const size_t n_pix = (size_t)gridDim.x * blockDim.x;            // words between the fields of generators and frames
    const size_t n_slots = n_pix * K;                               // words between the fields of the pixel / request state
    const size_t wave_index = (size_t)blockIdx.x * (blockDim.x / 64) + ww.wave;
    const size_t wave_base = wave_index * (64u * K);                // this wave's request slot 0
    const size_t gs = wave_index * 64u + ww.lane;                   // this lane's generator and frame stack
    const int N = A.cam.num_samples, sps = A.cam.sps;
    const float fs = (float)sps;
    const float eps = S.shadow_eps;
    uint32_t* ps_wave = A.pstate + wave_base;
    SlotMixedIO io{S, ps_wave, n_slots};
    const uint32_t n_lights = S.n_point_lights + A.n_area_lights;

    unsigned long long n_primary = 0, n_secondary = 0, n_shadow = 0;   // per lane
    bool drained = false;        // wave-uniform: the chunk's pixel cursor has passed the last pixel
    io.f(PS_IDX, ww.lane) = RT_SLOT_EMPTY;
    for (uint32_t k = 0; k < K; k++) io.f(PS_REQ, k * 64u + ww.lane) = OREQ_NONE;

#ifdef RT_ORDERED_TIMES
    unsigned long long tk_shade = 0, tk_trace = 0, tk_rounds = 0, tk_t0 = (unsigned long long)wall_clock64(), tk_init = 0, tk_iters = 0, tk_loop = 0, tk_active = 0;
#endif
    for (;;) {
#ifdef RT_ORDERED_TIMES
        const unsigned long long tk_a = (unsigned long long)wall_clock64();
#endif
        bool any_request = false, any_pixel = false;     // per lane
        {
            const uint32_t slot = ww.lane;                // the lane's request slot 0: pixel state, and the one request of a ray that continues the path
#define PSF(field) ps_wave[(size_t)(field) * n_slots + slot]
#define PSQ(field, q) ps_wave[(size_t)(field) * n_slots + (q) * 64u + ww.lane]      // request slot q of this lane
            MtGlobal gen;
            gen.mt = A.mt_scratch + gs;
            gen.stride = n_pix;
            uint32_t* perm = A.mt_scratch + 624 * n_pix + gs;
            float* fr = A.frames + gs;
#define OFR(depth, field) fr[((size_t)(depth) * OF_COUNT + (field)) * n_pix]
#define OFR_U(depth, field) (reinterpret_cast<uint32_t*>(fr)[((size_t)(depth) * OF_COUNT + (field)) * n_pix])
            bool lights_posted = false;                   // this round's requests are light tests, written to the slots directly

            uint32_t idx = PSF(PS_IDX);
            int px = 0, py = 0;
            int s = 0, depth = 0;
            V3 acc_color = mk(0.0f, 0.0f, 0.0f), ret = mk(0, 0, 0);
            bool ret_pending = false, start_sample = false;
            uint32_t hit_w0 = 0xFFFFFFFFu, hit_tri = 0;
            LaneTrace L;
            L.req = OREQ_NONE; L.has = false; L.t = 0.f; L.kind_index = 0xFFFFFFFFu; L.tri = 0; L.beta = L.gamma = 0.f;
            L.o = L.d = mk(0, 0, 0); L.time = 0.f; L.dist = 0.f;
            gen.idx = 624; gen.first = false; gen.lo = gen.hi = 0;

            // ---- an empty slot takes the next pixel of the chunk (one atomic per wave)
            if (!drained) {
                const uint64_t empty = __ballot(idx == RT_SLOT_EMPTY);
                if (empty != 0) {
                    uint32_t base = 0;
                    if (ww.lane == 0) base = atomicAdd(&ctl->cursor_closest[0], (uint32_t)__popcll(empty));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base + (uint32_t)__popcll(empty) >= A.g.n_idx) drained = true;
                    if (idx == RT_SLOT_EMPTY) {
                        const uint32_t cand_idx = base + (uint32_t)__popcll(empty & ((1ull << ww.lane) - 1ull));
                        const uint32_t mapped = (A.tile_order && cand_idx < A.g.n_idx) ? A.tile_order[cand_idx >> 6] * 64u + (cand_idx & 63u) : cand_idx;
                        if (cand_idx < A.g.n_idx && idx_to_pixel(A.g, mapped, px, py)) {
                            idx = mapped;
#ifdef RT_ORDERED_TIMES
                            const unsigned long long ti_a = (unsigned long long)wall_clock64();
#endif
                            gen.seed((uint32_t)(py * A.cam.image_w + px));
                            mt_shuffle_identity(gen, perm, n_pix, N);   // std::shuffle, as k_raygen_multi
                            start_sample = true;
#ifdef RT_ORDERED_TIMES
                            tk_init += (unsigned long long)wall_clock64() - ti_a;
#endif
                        }
                    } else {
                        (void)idx_to_pixel(A.g, idx, px, py);
                    }
                }
            }
            bool done = idx == RT_SLOT_EMPTY;
            if (!done && !start_sample) {
                // ---- the slot's pixel is under way: its place in the loop, and the answer to its request
                (void)idx_to_pixel(A.g, idx, px, py);
                const uint32_t fl = PSF(PS_FLAGS);
                ret_pending = (fl & 1u) != 0;
                s = (int)PSF(PS_S); depth = (int)PSF(PS_DEPTH);
                acc_color = mk(__uint_as_float(PSF(PS_ACC_R)), __uint_as_float(PSF(PS_ACC_G)), __uint_as_float(PSF(PS_ACC_B)));
                ret = mk(__uint_as_float(PSF(PS_RET_R)), __uint_as_float(PSF(PS_RET_G)), __uint_as_float(PSF(PS_RET_B)));
                hit_w0 = PSF(PS_HIT_W0); hit_tri = PSF(PS_HIT_TRI);
                gen.unpack(PSF(PS_GEN_IDX), PSF(PS_GEN_LO), PSF(PS_GEN_HI));
                L.time = __uint_as_float(PSF(PS_TIME));
                L.has = PSF(PS_HAS) != 0u; L.t = __uint_as_float(PSF(PS_T)); L.kind_index = PSF(PS_KIND); L.tri = PSF(PS_TRI);
                L.beta = __uint_as_float(PSF(PS_BETA)); L.gamma = __uint_as_float(PSF(PS_GAMMA));
            }

            // ---- run ApplyShading forward until it needs a ray traced (or the pixel is finished)
#ifdef RT_ORDERED_TIMES
            const unsigned long long tl_a = (unsigned long long)wall_clock64();
            unsigned tk_my_iters = 0;
#endif
            if (!done) {
                for (;;) {
#ifdef RT_ORDERED_TIMES
                    tk_my_iters++;
#endif
                    if (start_sample) {            // raytracer.cpp:56-74
                        start_sample = false;
                        const int sx = s % sps, sy = s / sps;
                        const float jx = (sx + gen.uniform()) / fs;
                        const float jy = (sy + gen.uniform()) / fs;
                        const int ai = (int)perm[(size_t)s * n_pix];
                        const int ax = ai % sps, ay = ai / sps;
                        const float au = (ax + gen.uniform()) / fs;
                        const float av = (ay + gen.uniform()) / fs;
                        L.time = gen.uniform();
                        V3 o, d;
                        compute_ray(A.cam, px, py, jx, jy, au, av, o, d);
                        depth = 0;
                        if (0 > S.max_depth) { ret = mk(0, 0, 0); ret_pending = true; depth = -1; }   // raytracer.cpp:384
                        else {
                            OFR(0, OF_OX) = o.x; OFR(0, OF_OY) = o.y; OFR(0, OF_OZ) = o.z;
                            OFR(0, OF_DX) = d.x; OFR(0, OF_DY) = d.y; OFR(0, OF_DZ) = d.z;
                            OFR_U(0, OF_STAGE) = OS_TRACE;
                            L.o = o; L.d = d; L.req = OREQ_CLOSEST;
                            n_primary++;
                            break;
                        }
                    }
                    if (ret_pending && depth < 0) {
                        // ComputeColor of the primary ray returned: raytracer.cpp:75-76
                        ret_pending = false;
                        acc_color = acc_color + ret;
                        s++;
                        if (s < N) { start_sample = true; continue; }
                        V3 c = acc_color * A.inv_n;                                       // raytracer.cpp:80-85
                        const int lx = px - A.g.x0, ly = py - A.g.y0;
                        uint8_t* o = A.rgb + (size_t)ly * A.row_stride + (size_t)lx * 3;
                        o[0] = (uint8_t)clampf(c.x, 0.0f, 255.0f);
                        o[1] = (uint8_t)clampf(c.y, 0.0f, 255.0f);
                        o[2] = (uint8_t)clampf(c.z, 0.0f, 255.0f);
                        if (A.hit_ids) {
                            uint32_t* ho = A.hit_ids + ((size_t)ly * A.g.w + lx) * 2;
                            ho[0] = hit_w0;
                            ho[1] = (hit_w0 != 0xFFFFFFFFu && (hit_w0 >> 28) == RT_KIND_MESH) ? A.tri_shade[hit_tri].face : (hit_w0 == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u);
                        }
                        done = true;
                        idx = RT_SLOT_EMPTY;       // (the slot takes its next pixel in the next round)
                        break;
                    }

                    // ---- the frame at `depth`
                    const uint32_t stage = OFR_U(depth, OF_STAGE);
                    const V3 o = mk(OFR(depth, OF_OX), OFR(depth, OF_OY), OFR(depth, OF_OZ));
                    const V3 d = mk(OFR(depth, OF_DX), OFR(depth, OF_DY), OFR(depth, OF_DZ));
                    bool to_lights = false, do_return = false, spawn = false, post_lights = false;
                    bool shaded = false;           // the frame has a hit: c / mat / color below are valid
                    V3 color = mk(0, 0, 0);
                    V3 child_o = mk(0, 0, 0), child_d = mk(0, 0, 0);
                    uint32_t light = 0;
                    ShadeCtx c;
                    c.type = RT_MAT_NONE;
                    uint32_t material_id = 1;

                    if (stage == OS_TRACE) {
                        // FindClosestHit returned (raytracer.cpp:387-400)
                        if (depth == 0 && s == 0) { hit_w0 = L.kind_index; hit_tri = L.tri; }
                        if (!L.has) {
                            ret = (depth == 0) ? ldv(S.background) : mk(0, 0, 0);
                            do_return = true;
                        } else {
                            const float tHit = L.t;
                            V3 hit_x = o + d * tHit;
                            int mid = 1;
                            V3 n = hit_normal(S, L.kind_index, L.tri, L.beta, L.gamma, hit_x, mid);
                            material_id = (uint32_t)mid;
                            OFR(depth, OF_THIT) = tHit;
                            OFR(depth, OF_NX) = n.x; OFR(depth, OF_NY) = n.y; OFR(depth, OF_NZ) = n.z;
                            OFR_U(depth, OF_MAT) = material_id;
                            shade_ctx(S, S.materials[material_id - 1], o, d, tHit, n, c);
                            shaded = true;
                            if (c.type == RT_MAT_MIRROR || c.type == RT_MAT_CONDUCTOR || c.type == RT_MAT_DIELECTRIC) {
                                // :1274-1276 / :1286-1291 / :1318-1319,:1333-1335 — the perturbation draws come first
                                child_d = perturb_reflection(c.wr, S.materials[material_id - 1].roughness, gen);
                                child_o = c.refl_origin;
                                spawn = true;
                                OFR_U(depth, OF_STAGE) = OS_REFLECT_WAIT;
                            } else {
                                to_lights = true;
                            }
                        }
                    } else {
                        const float tHit = OFR(depth, OF_THIT);
                        const V3 n = mk(OFR(depth, OF_NX), OFR(depth, OF_NY), OFR(depth, OF_NZ));
                        material_id = OFR_U(depth, OF_MAT);
                        shade_ctx(S, S.materials[material_id - 1], o, d, tHit, n, c);
                        shaded = true;
                        color = mk(OFR(depth, OF_CR), OFR(depth, OF_CG), OFR(depth, OF_CB));
                    }
                    const DMaterial& mat = S.materials[material_id - 1];
                    const V3 km = ldv(mat.km);

                    if (stage == OS_REFLECT_WAIT) {           // the reflected ray's ComputeColor returned
                        ret_pending = false;
                        if (c.type == RT_MAT_MIRROR) {
                            color = color + emul(km, ret);                                              // :1276-1277
                            to_lights = true;
                        } else if (c.type == RT_MAT_CONDUCTOR) {
                            V3 reflectColor = emul(km, ret);                                            // :1291-1292
                            color = color + reflectColor * c.Fr;
                            to_lights = true;
                        } else if (c.tir) {
                            color = color + emul(km, ret);                                              // :1319-1320
                            if (!c.entering) { ret = color; do_return = true; } else to_lights = true;  // :1364-1366
                        } else {
                            V3 reflectColor = emul(km, ret);                                            // :1335
                            OFR(depth, OF_RR) = reflectColor.x; OFR(depth, OF_RG) = reflectColor.y; OFR(depth, OF_RB) = reflectColor.z;
                            child_d = perturb_reflection(c.wt, mat.roughness, gen);                     // :1338-1340
                            child_o = c.refr_origin;
                            spawn = true;
                            OFR_U(depth, OF_STAGE) = OS_REFRACT_WAIT;
                        }
                    } else if (stage == OS_REFRACT_WAIT) {    // the refracted ray's ComputeColor returned (:1342-1361)
                        ret_pending = false;
                        V3 reflectColor = mk(OFR(depth, OF_RR), OFR(depth, OF_RG), OFR(depth, OF_RB));
                        V3 refractColor = ret;
                        if (!c.entering) {
                            float dd = vlen(c.x - o);
                            refractColor.x *= rt_expf(-mat.absorb_coef[0] * dd);
                            refractColor.y *= rt_expf(-mat.absorb_coef[1] * dd);
                            refractColor.z *= rt_expf(-mat.absorb_coef[2] * dd);
                        }
                        color = (color + reflectColor * c.Fr) + refractColor * (1.0f - c.Fr);
                        if (!c.entering) { ret = color; do_return = true; } else to_lights = true;
                    } else if (stage == OS_LIGHTS) {          // the answers of a batch of light tests, in the reference's order
                        light = OFR_U(depth, OF_LIGHT);
                        const uint32_t nl = min((uint32_t)K, n_lights - light);
                        for (uint32_t qq = 0; qq < nl; qq++, light++) {
                            const uint32_t q = nl - 1u - qq;  // (slots in reverse, see where they are posted)
                            const bool q_has = PSQ(PS_HAS, q) != 0u;
                            if (light < S.n_point_lights) {   // InShadow returned (:1371-1377)
                                if (!q_has) color = color + diffuse_specular(mat, S.point_lights[light], c.x, c.n_shading, c.w0);
                                continue;
                            }
                            // the closest hit towards the light sample returned (:1405-1411)
                            const DAreaLight& al = A.area_lights[light - S.n_point_lights];
                            const V3 wi = mk(__uint_as_float(PSQ(PS_DX, q)), __uint_as_float(PSQ(PS_DY, q)), __uint_as_float(PSQ(PS_DZ, q)));
                            const float dist_to_light = __uint_as_float(PSQ(PS_DIST, q));
                            bool in_sh = false;
                            if (q_has) {
                                const V3 so = c.x + c.n_shading * eps;
                                const V3 sh_x = so + wi * __uint_as_float(PSQ(PS_T, q));
                                const float dist_to_hit = vlen(sh_x - so);
                                if (dist_to_hit < dist_to_light - eps) in_sh = true;
                            }
                            if (!in_sh) {   // :1413-1431
                                float cos_light = fabsf(dot(mk(al.nx, al.ny, al.nz), wi * -1.0f));
                                float area = al.size * al.size;
                                float solid_angle = (area * cos_light) / (dist_to_light * dist_to_light);
                                V3 irradiance = mk(al.rx, al.ry, al.rz) * solid_angle;
                                float nd = dot(c.n_shading, wi);
                                float cos_theta = (0.0f < nd) ? nd : 0.0f;
                                V3 diffuse = emul(ldv(mat.kd), irradiance) * cos_theta;
                                V3 h = normalize(wi + c.w0);
                                float nh = dot(c.n_shading, h);
                                float cos_alpha = (0.0f < nh) ? nh : 0.0f;
                                float spec_factor = rt_powf(cos_alpha, mat.phong);
                                V3 specular = emul(ldv(mat.ks), irradiance) * spec_factor;
                                color = (color + diffuse) + specular;
                            }
                        }
                        post_lights = true;
                    }

                    if (to_lights) {
                        color = color + emul(ldv(S.ambient), ldv(mat.ka));           // :1368
                        light = 0;
                        post_lights = true;
                    }
                    // ---- the light loops (:1370-1433): post the next K tests — point lights first, then the area lights, whose
                    // two draws each are made here, in the reference's order — or finish the frame
                    if (post_lights) {
                        if (light < n_lights) {
                            const uint32_t nl = min((uint32_t)K, n_lights - light);
                            const V3 so = c.x + c.n_shading * eps;
                            // Slots are filled in reverse: the area lights' closest-hit rays, the long ones, get the low slots, which
                            // the wave's trace loop starts with; the any-hit rays towards the point lights fill lanes as they fall idle.
                            for (uint32_t qq = 0; qq < nl; qq++) {
                                const uint32_t j = light + qq, q = nl - 1u - qq;
                                V3 rd;
                                float rdist;
                                uint32_t kind;
                                if (j < S.n_point_lights) {
                                    const DPointLight Lp = S.point_lights[j];            // InShadow, :1008-1020
                                    V3 toLight = mk(Lp.px, Lp.py, Lp.pz) - c.x;
                                    rdist = vlen(toLight);
                                    rd = vdiv(toLight, rdist);
                                    kind = OREQ_ANYHIT;
                                } else {
                                    const DAreaLight& al = A.area_lights[j - S.n_point_lights];   // :1380-1403
                                    float u = (gen.uniform() - 0.5f) * al.size;
                                    float v = (gen.uniform() - 0.5f) * al.size;
                                    V3 tangent, bitangent;
                                    orthonormal_basis(mk(al.nx, al.ny, al.nz), tangent, bitangent);
                                    V3 sample = (mk(al.px, al.py, al.pz) + tangent * u) + bitangent * v;
                                    V3 to_light = sample - c.x;
                                    rdist = vlen(to_light);
                                    rd = vdiv(to_light, rdist);
                                    kind = OREQ_CLOSEST;
                                }
                                PSQ(PS_OX, q) = __float_as_uint(so.x); PSQ(PS_OY, q) = __float_as_uint(so.y); PSQ(PS_OZ, q) = __float_as_uint(so.z);
                                PSQ(PS_DX, q) = __float_as_uint(rd.x); PSQ(PS_DY, q) = __float_as_uint(rd.y); PSQ(PS_DZ, q) = __float_as_uint(rd.z);
                                PSQ(PS_DIST, q) = __float_as_uint(rdist);
                                PSQ(PS_REQ, q) = kind;
                                n_shadow++;
                            }
                            lights_posted = true;
                            OFR_U(depth, OF_STAGE) = OS_LIGHTS;
                            OFR_U(depth, OF_LIGHT) = light;
                        } else {
                            ret = color;
                            do_return = true;
                        }
                    }
                    if (shaded) { OFR(depth, OF_CR) = color.x; OFR(depth, OF_CG) = color.y; OFR(depth, OF_CB) = color.z; }

                    if (spawn) {
                        if (depth + 1 > S.max_depth) {            // ComputeColor's depth guard, raytracer.cpp:384-386: no ray
                            ret = mk(0, 0, 0);
                            ret_pending = true;
                            continue;
                        }
                        depth++;
                        OFR(depth, OF_OX) = child_o.x; OFR(depth, OF_OY) = child_o.y; OFR(depth, OF_OZ) = child_o.z;
                        OFR(depth, OF_DX) = child_d.x; OFR(depth, OF_DY) = child_d.y; OFR(depth, OF_DZ) = child_d.z;
                        OFR_U(depth, OF_STAGE) = OS_TRACE;
                        L.o = child_o; L.d = child_d; L.req = OREQ_CLOSEST;
                        n_secondary++;
                        break;
                    }
                    if (do_return) {
                        depth--;
                        ret_pending = true;
                        continue;
                    }
                    if (lights_posted) break;
                }
            }
#ifdef RT_ORDERED_TIMES
            {
                unsigned mx = tk_my_iters;
                for (int off = 32; off; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
                tk_iters += mx;
                tk_active += tk_my_iters;
                tk_loop += (unsigned long long)wall_clock64() - tl_a;
            }
#endif
            // ---- the slot's state goes back to memory
            PSF(PS_IDX) = idx;
            if (!done) {
                PSF(PS_FLAGS) = ret_pending ? 1u : 0u;
                PSF(PS_S) = (uint32_t)s; PSF(PS_DEPTH) = (uint32_t)depth;
                PSF(PS_ACC_R) = __float_as_uint(acc_color.x); PSF(PS_ACC_G) = __float_as_uint(acc_color.y); PSF(PS_ACC_B) = __float_as_uint(acc_color.z);
                PSF(PS_RET_R) = __float_as_uint(ret.x); PSF(PS_RET_G) = __float_as_uint(ret.y); PSF(PS_RET_B) = __float_as_uint(ret.z);
                PSF(PS_HIT_W0) = hit_w0; PSF(PS_HIT_TRI) = hit_tri;
                PSF(PS_GEN_IDX) = gen.packed_idx(); PSF(PS_GEN_LO) = gen.lo; PSF(PS_GEN_HI) = gen.hi;
                PSF(PS_TIME) = __float_as_uint(L.time);
                if (!lights_posted) {      // (a batch of light tests is in the slots already)
                    PSF(PS_OX) = __float_as_uint(L.o.x); PSF(PS_OY) = __float_as_uint(L.o.y); PSF(PS_OZ) = __float_as_uint(L.o.z);
                    PSF(PS_DX) = __float_as_uint(L.d.x); PSF(PS_DY) = __float_as_uint(L.d.y); PSF(PS_DZ) = __float_as_uint(L.d.z);
                    PSF(PS_DIST) = __float_as_uint(L.dist);
                }
                any_pixel = true;
            }
            if (!lights_posted) PSF(PS_REQ) = done ? (uint32_t)OREQ_NONE : (uint32_t)L.req;
            any_request = !done && (lights_posted || L.req != OREQ_NONE);
#undef OFR
#undef OFR_U
#undef PSF
#undef PSQ
        }
        // a slot that became empty in this round takes a new pixel in the next one, unless the chunk has none left
        if (!__any(any_pixel) && drained) break;
#ifdef RT_ORDERED_TIMES
        const unsigned long long tk_b = (unsigned long long)wall_clock64();
#endif

        // ---- the wave traces what its slots asked for: closest-hit and any-hit requests in one loop, 64 * SLOTS items
        // behind a cursor, so lanes whose ray is finished take the next request and long rays give subtrees away
        if (__any(any_request)) {
            if (ww.lane == 0) s_cursor[ww.wave] = 0;
            unsigned long long traced = 0;
            trace_persistent<2, false, NEAR>(S, io, 0u, 64u * K, &s_cursor[ww.wave], st, traced);
        }
#ifdef RT_ORDERED_TIMES
        { const unsigned long long tk_c = (unsigned long long)wall_clock64(); tk_shade += tk_b - tk_a; tk_trace += tk_c - tk_b; tk_rounds++; }
#endif
    }
#ifdef RT_ORDERED_TIMES
    if (ww.lane == 0 && (((blockIdx.x % 97) == 0 && ww.wave == 0) || ((unsigned long long)wall_clock64() - tk_t0) > 26000000ull))
        printf("[ordered] block %d: life %.0f us, %llu rounds: shade %.0f us (%.1f per round), trace %.0f us (%.1f per round) | lane 0: pixel init %.0f us | loop %.1f us per round, %.2f wave passes per round, lane 0 %.2f\n", (int)blockIdx.x,
               ((unsigned long long)wall_clock64() - tk_t0) * 0.01, tk_rounds, tk_shade * 0.01, tk_shade * 0.01 / tk_rounds, tk_trace * 0.01, tk_trace * 0.01 / tk_rounds, tk_init * 0.01,
               tk_loop * 0.01 / tk_rounds, (double)tk_iters / tk_rounds, (double)tk_active / tk_rounds);
#endif
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
#ifdef RT_TRACE_STATS
    atomicAdd(&ctl->dbg[0], acc.wave_node_steps); atomicAdd(&ctl->dbg[1], acc.lane_node_steps);
    atomicAdd(&ctl->dbg[2], acc.wave_tri_steps); atomicAdd(&ctl->dbg[3], acc.lane_tri_steps);
    atomicAdd(&ctl->dbg[4], acc.wave_outer); atomicAdd(&ctl->dbg[5], n_primary + n_secondary + n_shadow);
#endif
    if (n_primary) atomicAdd(&ctl->rays_primary, n_primary);
    if (n_secondary) atomicAdd(&ctl->rays_secondary, n_secondary);
    if (n_shadow) atomicAdd(&ctl->rays_shadow, n_shadow);
}
