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
    OF_STAGE, OF_LIGHT,                           // where ApplyShading stands
    OF_CR, OF_CG, OF_CB,                          // `color` so far
    OF_RR, OF_RG, OF_RB,                          // reflectColor (dielectric, raytracer.cpp:1335)
    OF_WX, OF_WY, OF_WZ, OF_DIST,                 // area light sample in flight: wi, dist_to_light
    OF_COUNT
};
enum { OS_TRACE = 0, OS_REFLECT_WAIT, OS_REFRACT_WAIT, OS_POINT, OS_AREA };
enum { OREQ_NONE = 0, OREQ_CLOSEST = 1, OREQ_ANYHIT = 2 };

// CreateOrthonormalBasis, raytracer.cpp:1522-1537
__device__ __forceinline__ void orthonormal_basis(V3 n, V3& tangent, V3& bitangent) {
    V3 helper;
    if (fabsf(n.x) < fabsf(n.y) && fabsf(n.x) < fabsf(n.z)) helper = mk(1.0f, 0.0f, 0.0f);
    else if (fabsf(n.y) < fabsf(n.z)) helper = mk(0.0f, 1.0f, 0.0f);
    else helper = mk(0.0f, 0.0f, 1.0f);
    tangent = normalize(cross(n, helper));
    bitangent = cross(n, tangent);
}
// PerturbReflection, raytracer.cpp:1438-1457
__device__ __forceinline__ V3 perturb_reflection(V3 r, float roughness, MtGlobal& gen) {
    if (roughness <= 0.0f) return r;
    V3 u, v;
    orthonormal_basis(r, u, v);
    const float xi1 = gen.uniform();
    const float xi2 = gen.uniform();
    V3 rr = (r + u * (roughness * (xi1 - 0.5f))) + v * (roughness * (xi2 - 0.5f));
    return normalize(rr);
}

// The request a lane has posted, and the answer it gets back.
struct LaneTrace {
    int req;                     // OREQ_*
    V3 o, d;
    float time, dist;            // dist: InShadow's minT = distance to the light
    bool has;                    // closest: hit found / any-hit: in shadow
    float t;
    uint32_t kind_index, tri;
    float beta, gamma;
    uint32_t bad_item;           // set if an item ever reaches a lane other than its own
};

struct LaneClosestIO {
    const DScene& S;
    LaneTrace& L;
    uint32_t lane;
    __device__ __forceinline__ int load(uint32_t i, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h) const {
        if (i != lane) { L.bad_item = 1; return 0; }
        if (L.req != OREQ_CLOSEST) return 0;
        o = L.o; d = L.d; time = L.time;
        minT = RT_FLT_MAX;
        has = false;
        h.kind_index = 0xFFFFFFFFu; h.tri = 0; h.beta = 0.f; h.gamma = 0.f;
        for (uint32_t k = 0; k < S.n_planes; k++) {   // raytracer.cpp:423-435
            float t;
            if (plane_test(o, d, S.planes[k], minT, t)) { minT = t; has = true; h.kind_index = (RT_KIND_PLANE << 28) | k; }
        }
        return 2;
    }
    __device__ __forceinline__ void store(uint32_t, bool has, const HitOut& h, float minT) const {
        L.has = has; L.t = minT; L.kind_index = has ? h.kind_index : 0xFFFFFFFFu; L.tri = h.tri; L.beta = h.beta; L.gamma = h.gamma;
        L.req = OREQ_NONE;
    }
};
struct LaneShadowIO {
    const DScene& S;
    LaneTrace& L;
    uint32_t lane;
    __device__ __forceinline__ int load(uint32_t i, V3& o, V3& d, float& time, float& minT, bool& has, HitOut&) const {
        if (i != lane) { L.bad_item = 1; return 0; }
        if (L.req != OREQ_ANYHIT) return 0;
        o = L.o; d = L.d; time = L.time;
        minT = L.dist;
        has = false;
        for (uint32_t k = 0; k < S.n_planes; k++) {   // raytracer.cpp:1022-1028
            float t;
            if (plane_test(o, d, S.planes[k], minT, t)) { L.has = true; L.req = OREQ_NONE; return 1; }
        }
        return 2;
    }
    __device__ __forceinline__ void store(uint32_t, bool shadow, const HitOut&, float) const { L.has = shadow; L.req = OREQ_NONE; }
};

// Both kinds in one trace loop (trace_persistent MODE 2): the lane's request says which its ray is.
struct LaneMixedIO {
    const DScene& S;
    LaneTrace& L;
    uint32_t lane;
    __device__ __forceinline__ int load(uint32_t i, V3& o, V3& d, float& time, float& minT, bool& has, HitOut& h, bool& any) const {
        if (i != lane) { L.bad_item = 1; return 0; }
        if (L.req == OREQ_CLOSEST) { any = false; return LaneClosestIO{S, L, lane}.load(i, o, d, time, minT, has, h); }
        if (L.req == OREQ_ANYHIT) { any = true; return LaneShadowIO{S, L, lane}.load(i, o, d, time, minT, has, h); }
        return 0;
    }
    __device__ __forceinline__ uint32_t first_blocks() const { return 1u; }     // the 64 lanes' requests, item = lane
    __device__ __forceinline__ uint32_t first_count() const { return 64u; }
    __device__ __forceinline__ void store(uint32_t i, bool verdict, const HitOut& h, float minT) const {
        if (L.req == OREQ_ANYHIT) LaneShadowIO{S, L, lane}.store(i, verdict, h, minT);
        else LaneClosestIO{S, L, lane}.store(i, verdict, h, minT);
    }
};

struct OrderedArgs {
    DCamera cam;
    ChunkGeom g;
    const DAreaLight* area_lights;
    uint32_t n_area_lights;
    uint32_t* mt_scratch;        // [624 + N][threads]
    float* frames;               // [max_depth + 1][OF_COUNT][threads]
    float inv_n;
    uint8_t* rgb;
    size_t row_stride;
    uint32_t* hit_ids;
    const DTriShade* tri_shade;
};

#ifndef RT_ORDERED_MIN_BLOCKS
#define RT_ORDERED_MIN_BLOCKS 3  // workgroups per CU the register budget is set for
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
    const size_t n_threads = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = A.cam.num_samples, sps = A.cam.sps;
    const float fs = (float)sps;
    const float eps = S.shadow_eps;
    MtGlobal gen;
    gen.mt = A.mt_scratch + tid;
    gen.stride = n_threads;
    gen.idx = 624;
    uint32_t* perm = A.mt_scratch + 624 * n_threads + tid;
    float* fr = A.frames + tid;
#define OFR(depth, field) fr[((size_t)(depth) * OF_COUNT + (field)) * n_threads]
#define OFR_U(depth, field) (reinterpret_cast<uint32_t*>(fr)[((size_t)(depth) * OF_COUNT + (field)) * n_threads])

    unsigned long long n_primary = 0, n_secondary = 0, n_shadow = 0;   // per lane
    LaneTrace L;
    L.req = OREQ_NONE; L.bad_item = 0; L.has = false; L.t = 0.f; L.kind_index = 0xFFFFFFFFu; L.tri = 0; L.beta = L.gamma = 0.f;
    L.o = L.d = mk(0, 0, 0); L.time = 0.f; L.dist = 0.f;

    // every wave walks the 256-pixel groups of the chunk with a grid stride; all 64 lanes stay in the loop
    const uint32_t n_groups = (A.g.n_idx + blockDim.x - 1) / blockDim.x;
    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const uint32_t idx = group * blockDim.x + threadIdx.x;
        int px = 0, py = 0;
        bool done = !(idx < A.g.n_idx && idx_to_pixel(A.g, idx, px, py));
        int s = 0, depth = 0;
        V3 acc_color = mk(0.0f, 0.0f, 0.0f);
        V3 ret = mk(0, 0, 0);
        bool ret_pending = false, start_sample = !done;
        uint32_t hit_w0 = 0xFFFFFFFFu, hit_tri = 0;
        if (!done) {
            gen.seed((uint32_t)(py * A.cam.image_w + px));
            for (int k = 0; k < N; k++) perm[(size_t)k * n_threads] = (uint32_t)k;
            int i = 1;   // std::shuffle, bits/stl_algo.h (pair path), as k_raygen_multi
            if ((N % 2) == 0) {
                uint32_t dpos = gen.lemire(2);
                uint32_t a = perm[(size_t)i * n_threads], b = perm[(size_t)dpos * n_threads];
                perm[(size_t)i * n_threads] = b; perm[(size_t)dpos * n_threads] = a;
                i++;
            }
            while (i < N) {
                uint32_t swap_range = (uint32_t)i + 1, b1 = swap_range + 1;
                uint32_t x = gen.lemire(swap_range * b1);
                uint32_t p0 = x / b1, p1 = x % b1;
                uint32_t a = perm[(size_t)i * n_threads], b = perm[(size_t)p0 * n_threads];
                perm[(size_t)i * n_threads] = b; perm[(size_t)p0 * n_threads] = a;
                i++;
                a = perm[(size_t)i * n_threads]; b = perm[(size_t)p1 * n_threads];
                perm[(size_t)i * n_threads] = b; perm[(size_t)p1 * n_threads] = a;
                i++;
            }
        }

        for (;;) {
            // ---- per lane: run ApplyShading forward until it needs a ray traced (or the pixel is finished)
            if (!done) {
                for (;;) {
                    if (start_sample) {            // raytracer.cpp:56-74
                        start_sample = false;
                        const int sx = s % sps, sy = s / sps;
                        const float jx = (sx + gen.uniform()) / fs;
                        const float jy = (sy + gen.uniform()) / fs;
                        const int ai = (int)perm[(size_t)s * n_threads];
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
                        break;
                    }

                    // ---- the frame at `depth`
                    const uint32_t stage = OFR_U(depth, OF_STAGE);
                    const V3 o = mk(OFR(depth, OF_OX), OFR(depth, OF_OY), OFR(depth, OF_OZ));
                    const V3 d = mk(OFR(depth, OF_DX), OFR(depth, OF_DY), OFR(depth, OF_DZ));
                    bool to_lights = false, do_return = false, spawn = false, next_point = false, next_area = false;
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
                    } else if (stage == OS_POINT) {           // InShadow returned (:1371-1377)
                        light = OFR_U(depth, OF_LIGHT);
                        if (!L.has) color = color + diffuse_specular(mat, S.point_lights[light], c.x, c.n_shading, c.w0);
                        light++;
                        next_point = true;
                    } else if (stage == OS_AREA) {            // the closest hit towards the light sample returned (:1405-1411)
                        light = OFR_U(depth, OF_LIGHT);
                        const DAreaLight& al = A.area_lights[light];
                        const V3 wi = mk(OFR(depth, OF_WX), OFR(depth, OF_WY), OFR(depth, OF_WZ));
                        const float dist_to_light = OFR(depth, OF_DIST);
                        bool in_sh = false;
                        if (L.has) {
                            const V3 so = c.x + c.n_shading * eps;
                            const V3 sh_x = so + wi * L.t;
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
                        light++;
                        next_area = true;
                    }

                    if (to_lights) {
                        color = color + emul(ldv(S.ambient), ldv(mat.ka));           // :1368
                        light = 0;
                        next_point = true;
                    }
                    // ---- the light loops: post the next test, or finish the frame
                    if (next_point) {
                        if (light < S.n_point_lights) {
                            const DPointLight Lp = S.point_lights[light];            // InShadow, :1008-1020
                            V3 toLight = mk(Lp.px, Lp.py, Lp.pz) - c.x;
                            float dist = vlen(toLight);
                            L.o = c.x + c.n_shading * eps;
                            L.d = vdiv(toLight, dist);
                            L.dist = dist;
                            L.req = OREQ_ANYHIT;
                            n_shadow++;
                            OFR_U(depth, OF_STAGE) = OS_POINT;
                            OFR_U(depth, OF_LIGHT) = light;
                        } else {
                            light = 0;
                            next_area = true;
                        }
                    }
                    if (next_area) {
                        if (light < A.n_area_lights) {
                            const DAreaLight& al = A.area_lights[light];              // :1380-1403
                            float u = (gen.uniform() - 0.5f) * al.size;
                            float v = (gen.uniform() - 0.5f) * al.size;
                            V3 tangent, bitangent;
                            orthonormal_basis(mk(al.nx, al.ny, al.nz), tangent, bitangent);
                            V3 sample = (mk(al.px, al.py, al.pz) + tangent * u) + bitangent * v;
                            V3 to_light = sample - c.x;
                            float dist_to_light = vlen(to_light);
                            V3 wi = vdiv(to_light, dist_to_light);
                            L.o = c.x + c.n_shading * eps;
                            L.d = wi;
                            L.req = OREQ_CLOSEST;
                            n_shadow++;
                            OFR_U(depth, OF_STAGE) = OS_AREA;
                            OFR_U(depth, OF_LIGHT) = light;
                            OFR(depth, OF_WX) = wi.x; OFR(depth, OF_WY) = wi.y; OFR(depth, OF_WZ) = wi.z;
                            OFR(depth, OF_DIST) = dist_to_light;
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
                    if (L.req != OREQ_NONE) break;
                }
            }
            if (!__any(!done)) break;

            // ---- the wave traces what its lanes asked for
#ifdef RT_ORDERED_TWO_LOOPS
            if (__any(L.req == OREQ_CLOSEST)) {
                if (ww.lane == 0) s_cursor[ww.wave] = 0;
                LaneClosestIO io{S, L, ww.lane};
                unsigned long long traced = 0;
                trace_persistent<0, false, NEAR>(S, io, 0u, 64u, &s_cursor[ww.wave], st, traced);
            }
            if (__any(L.req == OREQ_ANYHIT)) {
                if (ww.lane == 0) s_cursor[ww.wave] = 0;
                LaneShadowIO io{S, L, ww.lane};
                unsigned long long traced = 0;
                trace_persistent<1, false, false>(S, io, 0u, 64u, &s_cursor[ww.wave], st, traced);
            }
#else
            // closest-hit and any-hit requests share one loop: all lanes with a request are busy in it together,
            // and the kernel holds one copy of the traversal instead of two
            if (__any(L.req != OREQ_NONE)) {
                if (ww.lane == 0) s_cursor[ww.wave] = 0;
                LaneMixedIO io{S, L, ww.lane};
                unsigned long long traced = 0;
                trace_persistent<2, false, NEAR>(S, io, 0u, 64u, &s_cursor[ww.wave], st, traced);
            }
#endif
        }
    }
#undef OFR
#undef OFR_U
    if (st.overflow) atomicOr(&ctl->overflow, (st.overflow & 48u) | ((st.overflow & ~48u) ? 4u : 0u));
    if (L.bad_item) atomicOr(&ctl->overflow, 8u);
#ifdef RT_TRACE_STATS
    atomicAdd(&ctl->dbg[0], acc.wave_node_steps); atomicAdd(&ctl->dbg[1], acc.lane_node_steps);
    atomicAdd(&ctl->dbg[2], acc.wave_tri_steps); atomicAdd(&ctl->dbg[3], acc.lane_tri_steps);
    atomicAdd(&ctl->dbg[4], acc.wave_outer); atomicAdd(&ctl->dbg[5], n_primary + n_secondary + n_shadow);
#endif
    if (n_primary) atomicAdd(&ctl->rays_primary, n_primary);
    if (n_secondary) atomicAdd(&ctl->rays_secondary, n_secondary);
    if (n_shadow) atomicAdd(&ctl->rays_shadow, n_shadow);
}
