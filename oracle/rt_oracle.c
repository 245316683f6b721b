/*
 * rt_oracle.c — TEST INFRASTRUCTURE.  CPU restatement of the reference's
 * per-pixel hot path (fatih-ozdal/Raytracer, raytracer.cpp:40-87 and everything
 * it calls) over the flattened scene of include/rt.h.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product (librt_hip.so) never
 * links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks that this
 * file reproduces, byte for byte, (a) the reference's own committed renders
 * blog/hw3/my_outputs/{spheres_dof,cornellbox_area,cornellbox_brushed_metal,
 * metal_glass_plates,cornellbox_boxes_dynamic}.png, from_hw2/{mirror_room,
 * metal_glass_plates,two_berserkers}.png, tap_water/tap_0000.png (stored as
 * fixtures under tests/golden/), and (b) the output of the reference itself
 * compiled from /root/reference (oracle/_ref/libref_oracle.so) when present.
 *
 * Third-party arithmetic the reference path depends on and that is NOT under
 * /root/reference (restated from the published algorithms):
 *   - libstdc++ 11 std::mt19937 (MT19937, Matsumoto & Nishimura 1998),
 *     std::shuffle (bits/stl_algo.h: pairwise swap positions from one draw),
 *     std::uniform_int_distribution (Lemire 2019 nearly-divisionless,
 *     bits/uniform_int_dist.h _S_nd), std::generate_canonical<float,24>
 *     (bits/random.tcc: float(u32) / 2^32, clamped to nextafter(1,0)).
 *   - glibc 2.35 libm powf / expf / sqrtf: called as-is (same libm as the
 *     reference build on this image).
 *
 * Arithmetic rules: IEEE binary32, no FMA contraction (-ffp-contract=off; the
 * reference is built for baseline x86-64, which has no FMA), operation order
 * exactly as in the cited reference lines.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "rt.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define RT_FLT_MAX 3.402823466e+38F /* MathF.h:6 */
#define EPS_PARALLEL 1e-8f          /* raytracer.h:12 */
#define RAY_MISS -1.0f              /* raytracer.h:13 */

/* ------------------------------------------------------------------ counters */
typedef struct rto_counters {
    /* closest-hit rays (FindClosestHit) */
    uint64_t c_rays, c_top_nodes, c_mesh_nodes, c_tri_tests, c_sphere_tests, c_plane_tests, c_inst_entries;
    /* shadow rays (InShadow) */
    uint64_t s_rays, s_top_nodes, s_mesh_nodes, s_tri_tests, s_sphere_tests, s_plane_tests, s_inst_entries;
    uint64_t rays_primary, rays_secondary;
    uint64_t max_stack;   /* deepest traversal stack seen (entries) */
} rto_counters;

typedef struct {
    uint64_t rays, top_nodes, mesh_nodes, tri_tests, sphere_tests, plane_tests, inst_entries;
} ray_ctr;

typedef struct {
    ray_ctr c, s;
    uint64_t rays_primary, rays_secondary, max_stack;
} thread_ctr;

/* ------------------------------------------------------------------ Vec3f.h */
typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vld(const float* p) { return V(p[0], p[1], p[2]); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              /* Vec3f.h:16-18 */
static inline v3 cross(v3 a, v3 b) {                                                             /* Vec3f.h:20-24 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlen(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }             /* Vec3f.h:26-28 */
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 emul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }

/* MathF.h:18-28 */
static inline float q_rsqrt(float x) {
    if (x <= 0.0f) return 0.0f;
    uint32_t i;
    memcpy(&i, &x, 4);
    i = 0x5f3759dfu - (i >> 1);
    float y;
    memcpy(&y, &i, 4);
    y = y * (1.5f - 0.5f * x * y * y);
    return y;
}
/* Vec3f.h:30-39 */
static inline v3 normalize(v3 a) {
    float len2 = a.x * a.x + a.y * a.y + a.z * a.z;
    float inv = q_rsqrt(len2);
    if (inv == 0.0f) return V(0.f, 0.f, 0.f);
    return mul(a, inv);
}
/* MathF.h:30-38 */
static inline float det3x3(float a11, float a12, float a13, float a21, float a22, float a23, float a31, float a32,
                           float a33) {
    return a11 * (a22 * a33 - a23 * a32) - a12 * (a21 * a33 - a23 * a31) + a13 * (a21 * a32 - a22 * a31);
}
static inline float clampf(float v, float lo, float hi) { /* MathF.h:11-16 */
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
/* Mat4f.h:21-27 (m row-major) */
static inline v3 xf_point(const float* m, v3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1;
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1;
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1;
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * 1;
    return V(x / w, y / w, z / w);
}
/* Mat4f.h:30-36 */
static inline v3 xf_vector(const float* m, v3 v) {
    float x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 0;
    float y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 0;
    float z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 0;
    return V(x, y, z);
}
/* invTransformation.transpose().transformVector(n): Mat4f.h:51-59 + 30-36 */
static inline v3 xf_vector_transposed(const float* m, v3 v) {
    float x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * 0;
    float y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * 0;
    float z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * 0;
    return V(x, y, z);
}
/* Mat4f.h:126-130 */
static inline float mat_det3(const float* m) {
    return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
           m[2] * (m[4] * m[9] - m[5] * m[8]);
}

/* --------------------------------------------------- libstdc++ RNG restated */
typedef struct { uint32_t mt[624]; int idx; } mt19937;
static void mt_seed(mt19937* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static void mt_twist(mt19937* g) {
    uint32_t* mt = g->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}
static inline uint32_t mt_next(mt19937* g) {
    if (g->idx >= 624) mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
/* bits/uniform_int_dist.h _S_nd<uint64_t>(g, range) */
static uint32_t lemire(mt19937* g, uint32_t range) {
    uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}
/* bits/stl_algo.h std::shuffle, pair path (n*n <= 2^32-1) and the generic tail */
static void shuffle_ints(int* a, int n, mt19937* g) {
    if (n <= 0) return;
    uint64_t urange = (uint64_t)n;
    if (0xFFFFFFFFull / urange >= urange) {
        int i = 1;
        if ((n % 2) == 0) {
            uint32_t d = lemire(g, 2);
            int t = a[i]; a[i] = a[d]; a[d] = t;
            i++;
        }
        while (i != n) {
            uint32_t swap_range = (uint32_t)i + 1;
            uint32_t b1 = swap_range + 1;
            uint32_t x = lemire(g, swap_range * b1);
            uint32_t p0 = x / b1, p1 = x % b1;
            int t = a[i]; a[i] = a[p0]; a[p0] = t; i++;
            t = a[i]; a[i] = a[p1]; a[p1] = t; i++;
        }
        return;
    }
    for (int i = 1; i < n; i++) {
        uint32_t p = lemire(g, (uint32_t)i + 1);
        int t = a[i]; a[i] = a[p]; a[p] = t;
    }
}
/* generate_canonical<float,24> + uniform_real_distribution(0,1) */
static inline float mt_uniform(mt19937* g) {
    float r = (float)mt_next(g) / 4294967296.0f;
    if (r >= 1.0f) r = 0x1.fffffep-1f;
    return r * 1.0f + 0.0f;
}

/* ------------------------------------------------------------------- types */
typedef struct { v3 o, d; int depth; float time; } ray_t; /* raytracer.h:15-20 */

typedef struct {
    const rt_scene_desc* sc;
    const rt_camera_desc* cam;
    thread_ctr* ctr;
} ctx_t;

typedef struct {
    int kind;          /* RT_KIND_* */
    int index;         /* object index */
    int material_id;
    uint32_t face_idx; /* mesh: face index inside the mesh's face set */
    rt_face face;      /* mesh / triangle */
    int smooth;
    float beta, gamma;
} hit_info;

/* ----------------------------------------------------- primitive tests */
/* IntersectAABB, raytracer.cpp:815-894 */
static float intersect_aabb(const ray_t* r, const float* bmin, const float* bmax, float minT) {
    float tMin = 0.0f, tMax = minT;
    const float o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
    for (int a = 0; a < 3; a++) {
        if (d[a] != 0.0f) {
            float invD = 1.0f / d[a];
            float t0 = (bmin[a] - o[a]) * invD;
            float t1 = (bmax[a] - o[a]) * invD;
            if (t1 < t0) { float t = t0; t0 = t1; t1 = t; }
            tMin = (tMin < t0) ? t0 : tMin; /* std::max(tMin,t0) */
            tMax = (t1 < tMax) ? t1 : tMax; /* std::min(tMax,t1) */
            if (tMax < tMin) return RAY_MISS;
        } else {
            if (o[a] < bmin[a] || o[a] > bmax[a]) return RAY_MISS;
        }
    }
    float tHit;
    if (tMin > 0.0f) tHit = tMin;
    else {
        if (tMax <= 0.0f) return RAY_MISS;
        tHit = tMax;
    }
    if (tHit > minT) return RAY_MISS;
    return tHit;
}

/* IntersectsTriangle_Bary, raytracer.cpp:896-947 */
static float intersect_triangle(const ray_t* r, const rt_face* f, const rt_vertex* vd, float minT, float* beta_out,
                                float* gamma_out) {
    v3 va = vld(vd[f->i0 - 1].pos), vb = vld(vd[f->i1 - 1].pos), vc = vld(vd[f->i2 - 1].pos);
    v3 A0 = sub(va, vb), A1 = sub(va, vc), A2 = r->d, b = sub(va, r->o);
    float detA = det3x3(A0.x, A1.x, A2.x, A0.y, A1.y, A2.y, A0.z, A1.z, A2.z);
    if (fabsf(detA) < EPS_PARALLEL) return RAY_MISS;
    float beta = det3x3(b.x, A1.x, A2.x, b.y, A1.y, A2.y, b.z, A1.z, A2.z) / detA;
    if (beta < -EPS_PARALLEL) return RAY_MISS;
    float gamma = det3x3(A0.x, b.x, A2.x, A0.y, b.y, A2.y, A0.z, b.z, A2.z) / detA;
    if ((gamma < -EPS_PARALLEL) || ((beta + gamma) > (1.0f + EPS_PARALLEL))) return RAY_MISS;
    float t = det3x3(A0.x, A1.x, b.x, A0.y, A1.y, b.y, A0.z, A1.z, b.z) / detA;
    if (t < -EPS_PARALLEL || t >= minT) return RAY_MISS;
    *beta_out = beta;
    *gamma_out = gamma;
    return t;
}

/* IntersectSphere, raytracer.cpp:949-980 (sqrt(double) narrowed == sqrtf) */
static float intersect_sphere(const ray_t* r, v3 center, float radius, float minT) {
    v3 oc = sub(r->o, center), d = r->d;
    float A = dot(d, d);
    float B = 2 * dot(d, oc);
    float C = dot(oc, oc) - radius * radius;
    float delta = B * B - 4 * A * C;
    if (delta < 0.0f) return RAY_MISS;
    float sq = (float)sqrt((double)delta);
    float t1 = (-B - sq) / (2 * A);
    float t2 = (-B + sq) / (2 * A);
    float t;
    if (t1 > 0.0f) t = t1;
    else if (t2 > 0.0f) t = t2;
    else return RAY_MISS;
    if (t >= minT) return RAY_MISS;
    return t;
}

/* IntersectsPlane, raytracer.cpp:982-999 */
static float intersect_plane(const ray_t* r, v3 n, float plane_d, float minT) {
    float den = dot(r->d, n);
    if (fabsf(den) < EPS_PARALLEL) return RAY_MISS;
    float t = -(dot(r->o, n) + plane_d) / den;
    if (t < 0.0f || t >= minT) return RAY_MISS;
    return t;
}

/* IntersectMeshBVH, raytracer.cpp:758-811 */
static float intersect_mesh_bvh(const ctx_t* cx, ray_ctr* rc, const ray_t* r, int bvh_index, float minT,
                                uint32_t* face_idx_out, float* beta_out, float* gamma_out, uint32_t stack_base) {
    const rt_scene_desc* sc = cx->sc;
    const rt_mesh_bvh* bvh = &sc->mesh_bvhs[bvh_index];
    if (bvh->node_count == 0) return RAY_MISS; /* empty mesh: nothing to hit */
    const rt_bvh_node* nodes = sc->mesh_nodes + bvh->node_offset;
    const uint32_t* prim_face = sc->mesh_prim_face + bvh->prim_offset;
    const rt_face* faces = sc->faces + bvh->face_offset;
    float meshMinT = minT;
    int found = 0;
    uint32_t stack[64];
    uint32_t sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const rt_bvh_node* node = &nodes[stack[--sp]];
        rc->mesh_nodes++;
        if (intersect_aabb(r, node->bmin, node->bmax, meshMinT) == RAY_MISS) continue;
        if (node->prim_count > 0) {
            for (int i = 0; i < node->prim_count; i++) {
                uint32_t fi = prim_face[node->left_or_first + i];
                float tb, tg;
                rc->tri_tests++;
                float t = intersect_triangle(r, &faces[fi], sc->vertex_data, meshMinT, &tb, &tg);
                if (t < meshMinT && t != RAY_MISS) {
                    meshMinT = t;
                    *face_idx_out = fi;
                    *beta_out = tb;
                    *gamma_out = tg;
                    found = 1;
                }
            }
        } else {
            if (sp + 2 > 64) abort(); /* the reference's stack[64] would overflow here too */
            stack[sp++] = (uint32_t)node->left_or_first + 1;
            stack[sp++] = (uint32_t)node->left_or_first;
            if (stack_base + sp > cx->ctr->max_stack) cx->ctr->max_stack = stack_base + sp;
        }
    }
    return found ? meshMinT : RAY_MISS;
}

/* The per-object ray set-up shared by the three cases of IntersectTopBVH
 * (raytracer.cpp:559-587, 627-655, 686-714) and of InShadow (:1070-1095 ...). */
static inline void object_space_ray(const ray_t* ray, uint32_t flags, const float* inv, const float* mb, float minT,
                                    ray_t* test, float* testMinT, ray_ctr* rc) {
    *test = *ray;
    *testMinT = minT;
    if (flags & RT_OBJ_HAS_TRANSFORM) {
        v3 ro = ray->o, rd = ray->d;
        if (flags & RT_OBJ_HAS_MOTION_BLUR) ro = sub(ro, mul(vld(mb), ray->time));
        test->o = xf_point(inv, ro);
        test->d = normalize(xf_vector(inv, rd));
        v3 objDir = xf_vector(inv, ray->d);
        float scale = vlen(objDir);
        *testMinT = minT * scale;
        rc->inst_entries++;
    } else if (flags & RT_OBJ_HAS_MOTION_BLUR) {
        test->o = sub(ray->o, mul(vld(mb), ray->time));
    }
}
/* raytracer.cpp:598-606 */
static inline float world_t(const ray_t* ray, const ray_t* test, float t, uint32_t flags, const float* M,
                            const float* mb) {
    float worldT = t;
    if (flags & RT_OBJ_HAS_TRANSFORM) {
        v3 objHit = add(test->o, mul(test->d, t));
        v3 worldHit = xf_point(M, objHit);
        if (flags & RT_OBJ_HAS_MOTION_BLUR) worldHit = add(worldHit, mul(vld(mb), ray->time));
        worldT = vlen(sub(worldHit, ray->o));
    }
    return worldT;
}

/* IntersectTopBVH (raytracer.cpp:527-756) when any_hit == 0;
 * the traversal of InShadow (raytracer.cpp:1030-1243) when any_hit == 1
 * (returns 1 at the first accepted hit). */
static int traverse_top(const ctx_t* cx, ray_ctr* rc, const ray_t* ray, float* minT, int* has, hit_info* hi,
                        int any_hit) {
    const rt_scene_desc* sc = cx->sc;
    if (sc->n_top_nodes == 0) return 0;
    uint32_t stack[64];
    uint32_t sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const rt_bvh_node* node = &sc->top_nodes[stack[--sp]];
        rc->top_nodes++;
        if (intersect_aabb(ray, node->bmin, node->bmax, *minT) == RAY_MISS) continue;
        if (node->prim_count > 0) {
            for (int i = 0; i < node->prim_count; i++) {
                const rt_top_prim* prim = &sc->top_prims[sc->top_prim_idx[node->left_or_first + i]];
                ray_t test;
                float testMinT;
                switch (prim->kind) {
                case RT_KIND_MESH: {
                    const rt_mesh* m = &sc->meshes[prim->index];
                    object_space_ray(ray, m->flags, m->inv_transform, m->motion_blur, *minT, &test, &testMinT, rc);
                    uint32_t fi = 0;
                    float tb = 0, tg = 0;
                    float t = intersect_mesh_bvh(cx, rc, &test, m->bvh_index, testMinT, &fi, &tb, &tg, sp);
                    if (t < testMinT && t != RAY_MISS) {
                        float wt = world_t(ray, &test, t, m->flags, m->transform, m->motion_blur);
                        if (wt < *minT) {
                            if (any_hit) return 1;
                            *minT = wt;
                            hi->material_id = m->material_id;
                            hi->kind = RT_KIND_MESH;
                            hi->face_idx = fi;
                            hi->face = sc->faces[sc->mesh_bvhs[m->bvh_index].face_offset + fi];
                            hi->smooth = (m->flags & RT_OBJ_SMOOTH) != 0;
                            hi->beta = tb;
                            hi->gamma = tg;
                            hi->index = (int)prim->index;
                            *has = 1;
                        }
                    }
                    break;
                }
                case RT_KIND_SPHERE: {
                    const rt_sphere* s = &sc->spheres[prim->index];
                    object_space_ray(ray, s->flags, s->inv_transform, s->motion_blur, *minT, &test, &testMinT, rc);
                    rc->sphere_tests++;
                    float t = intersect_sphere(&test, vld(sc->vertex_data[s->center_vertex_id - 1].pos), s->radius,
                                               testMinT);
                    if (t < testMinT && t != RAY_MISS) {
                        float wt = world_t(ray, &test, t, s->flags, s->transform, s->motion_blur);
                        if (wt < *minT) {
                            if (any_hit) return 1;
                            *minT = wt;
                            hi->material_id = s->material_id;
                            hi->kind = RT_KIND_SPHERE;
                            hi->index = (int)prim->index;
                            *has = 1;
                        }
                    }
                    break;
                }
                case RT_KIND_TRIANGLE: {
                    const rt_triangle* tr = &sc->triangles[prim->index];
                    object_space_ray(ray, tr->flags, tr->inv_transform, tr->motion_blur, *minT, &test, &testMinT, rc);
                    float db, dg;
                    rc->tri_tests++;
                    float t = intersect_triangle(&test, &tr->face, sc->vertex_data, testMinT, &db, &dg);
                    if (t < testMinT && t != RAY_MISS) {
                        float wt = world_t(ray, &test, t, tr->flags, tr->transform, tr->motion_blur);
                        if (wt < *minT) {
                            if (any_hit) return 1;
                            *minT = wt;
                            hi->material_id = tr->material_id;
                            hi->kind = RT_KIND_TRIANGLE;
                            hi->face = tr->face;
                            hi->face_idx = 0;
                            hi->index = (int)prim->index;
                            *has = 1;
                        }
                    }
                    break;
                }
                default: break;
                }
            }
        } else {
            if (sp + 2 > 64) abort();
            stack[sp++] = (uint32_t)node->left_or_first + 1;
            stack[sp++] = (uint32_t)node->left_or_first;
            if (sp > cx->ctr->max_stack) cx->ctr->max_stack = sp;
        }
    }
    return 0;
}

typedef struct { int material_id; v3 x, n; int kind; } hit_record; /* HitRecord.h:25-31 */

/* FindClosestHit, raytracer.cpp:405-525 */
static int find_closest_hit(const ctx_t* cx, const ray_t* ray, hit_record* out, hit_info* hi_out) {
    const rt_scene_desc* sc = cx->sc;
    ray_ctr* rc = &cx->ctr->c;
    rc->rays++;
    float minT = RT_FLT_MAX;
    int has = 0;
    hit_info hi;
    memset(&hi, 0, sizeof(hi));
    for (uint32_t i = 0; i < sc->n_planes; i++) {
        const rt_plane* p = &sc->planes[i];
        rc->plane_tests++;
        float t = intersect_plane(ray, vld(p->n_unit), p->plane_d, minT);
        if (t < minT && t != RAY_MISS) {
            minT = t;
            hi.material_id = p->material_id;
            hi.kind = RT_KIND_PLANE;
            hi.index = (int)i;
            has = 1;
        }
    }
    traverse_top(cx, rc, ray, &minT, &has, &hi, 0);
    if (hi_out) *hi_out = hi;
    if (!has) return 0;

    v3 hit_x = add(ray->o, mul(ray->d, minT));
    v3 normal;
    switch (hi.kind) {
    case RT_KIND_MESH: {
        const rt_mesh* m = &sc->meshes[hi.index];
        v3 objN;
        if (hi.smooth) {
            v3 nA = vld(sc->vertex_data[hi.face.i0 - 1].normal);
            v3 nB = vld(sc->vertex_data[hi.face.i1 - 1].normal);
            v3 nC = vld(sc->vertex_data[hi.face.i2 - 1].normal);
            float alpha = 1.f - hi.beta - hi.gamma;
            objN = normalize(add(add(mul(nA, alpha), mul(nB, hi.beta)), mul(nC, hi.gamma)));
        } else {
            objN = vld(hi.face.n_unit);
        }
        int xf = (m->flags & RT_OBJ_HAS_TRANSFORM) != 0;
        if (xf && mat_det3(m->transform) < 0.0f) objN = mul(objN, -1.0f);
        normal = xf ? normalize(xf_vector_transposed(m->inv_transform, objN)) : objN;
        break;
    }
    case RT_KIND_TRIANGLE: {
        const rt_triangle* tr = &sc->triangles[hi.index];
        v3 objN = vld(hi.face.n_unit);
        int xf = (tr->flags & RT_OBJ_HAS_TRANSFORM) != 0;
        normal = xf ? normalize(xf_vector_transposed(tr->inv_transform, objN)) : objN;
        if (xf && mat_det3(tr->transform) < 0.0f) normal = mul(normal, -1.0f);
        break;
    }
    case RT_KIND_SPHERE: {
        const rt_sphere* s = &sc->spheres[hi.index];
        v3 c = vld(sc->vertex_data[s->center_vertex_id - 1].pos);
        int xf = (s->flags & RT_OBJ_HAS_TRANSFORM) != 0;
        if (xf) {
            v3 objHit = xf_point(s->inv_transform, hit_x);
            v3 objN = normalize(sub(objHit, c));
            normal = normalize(xf_vector_transposed(s->inv_transform, objN));
        } else {
            normal = normalize(sub(hit_x, c));
        }
        if (xf && mat_det3(s->transform) < 0.0f) normal = mul(normal, -1.0f);
        break;
    }
    case RT_KIND_PLANE: normal = vld(sc->planes[hi.index].n_unit); break;
    default: normal = V(0, 0, 0);
    }
    out->material_id = hi.material_id;
    out->x = hit_x;
    out->n = normal;
    out->kind = hi.kind;
    return 1;
}

/* InShadow, raytracer.cpp:1008-1244 */
static int in_shadow(const ctx_t* cx, v3 point, const rt_point_light* L, v3 n, float eps, float time) {
    const rt_scene_desc* sc = cx->sc;
    ray_ctr* rc = &cx->ctr->s;
    rc->rays++;
    v3 toLight = sub(vld(L->position), point);
    float dist = vlen(toLight);
    ray_t sr;
    sr.o = add(point, mul(n, eps));
    sr.d = divs(toLight, dist);
    sr.depth = 0;
    sr.time = time;
    float minT = dist;
    for (uint32_t i = 0; i < sc->n_planes; i++) {
        rc->plane_tests++;
        float t = intersect_plane(&sr, vld(sc->planes[i].n_unit), sc->planes[i].plane_d, minT);
        if (t < minT && t != RAY_MISS) return 1;
    }
    int has = 0;
    hit_info hi;
    return traverse_top(cx, rc, &sr, &minT, &has, &hi, 1);
}

/* CreateOrthonormalBasis, raytracer.cpp:1522-1537 */
static void orthonormal_basis(v3 n, v3* tangent, v3* bitangent) {
    v3 helper;
    if (fabsf(n.x) < fabsf(n.y) && fabsf(n.x) < fabsf(n.z)) helper = V(1.0f, 0.0f, 0.0f);
    else if (fabsf(n.y) < fabsf(n.z)) helper = V(0.0f, 1.0f, 0.0f);
    else helper = V(0.0f, 0.0f, 1.0f);
    *tangent = normalize(cross(n, helper));
    *bitangent = cross(n, *tangent);
}
/* PerturbReflection, raytracer.cpp:1438-1457 */
static v3 perturb(v3 r, float roughness, mt19937* g) {
    if (roughness <= 0.0f) return r;
    v3 u, v;
    orthonormal_basis(r, &u, &v);
    float xi1 = mt_uniform(g);
    float xi2 = mt_uniform(g);
    v3 rr = add(add(r, mul(u, roughness * (xi1 - 0.5f))), mul(v, roughness * (xi2 - 0.5f)));
    return normalize(rr);
}
/* raytracer.cpp:1459-1470 */
static float fresnel_dielectric(float cosTheta, float cosPhi, float n1, float n2) {
    float r_par = (n2 * cosTheta - n1 * cosPhi) / (n2 * cosTheta + n1 * cosPhi);
    float r_perp = (n1 * cosTheta - n2 * cosPhi) / (n1 * cosTheta + n2 * cosPhi);
    return (r_par * r_par + r_perp * r_perp) * 0.5f;
}
/* raytracer.cpp:1472-1488 */
static float fresnel_conductor(float cosTheta, float n, float k) {
    float cos_sq = cosTheta * cosTheta;
    float n_sq = n * n;
    float k_sq = k * k;
    float two_n_cos = 2.0f * n * cosTheta;
    float rs_num = (n_sq + k_sq) - two_n_cos + cos_sq;
    float rs_den = (n_sq + k_sq) + two_n_cos + cos_sq;
    float R_S = rs_num / rs_den;
    float rp_num = (n_sq + k_sq) * cos_sq - two_n_cos + 1;
    float rp_den = (n_sq + k_sq) * cos_sq + two_n_cos + 1;
    float R_P = rp_num / rp_den;
    return (R_S + R_P) * 0.5f;
}
/* ComputeDiffuseAndSpecular, raytracer.cpp:1490-1520 */
static v3 diffuse_specular(const rt_material* mat, const rt_point_light* light, v3 point, v3 normal, v3 w0) {
    v3 L = sub(vld(light->position), point);
    v3 wi = normalize(L);
    float cos_theta = dot(wi, normal);
    if (cos_theta < 0.0f) return V(0.0f, 0.0f, 0.f);
    float d_sq = dot(L, L);
    v3 irradiance = divs(vld(light->intensity), d_sq);
    v3 diffuse = emul(mul(vld(mat->diffuse_refl), cos_theta), irradiance);
    v3 h = normalize(add(wi, w0));
    float cos_alpha = dot(normal, h);
    if (cos_alpha < 0.0f) return diffuse;
    v3 spec = emul(mul(vld(mat->specular_refl), powf(cos_alpha, mat->phong_exponent)), irradiance);
    return add(diffuse, spec);
}

static v3 compute_color(const ctx_t* cx, const ray_t* ray, mt19937* g);

static v3 child_color(const ctx_t* cx, const ray_t* parent, v3 origin, v3 dir, mt19937* g) {
    ray_t r;
    r.o = origin;
    r.d = dir;
    r.depth = parent->depth + 1;
    r.time = parent->time;
    return compute_color(cx, &r, g);
}

/* ApplyShading, raytracer.cpp:1248-1436 */
static v3 apply_shading(const ctx_t* cx, const ray_t* ray, const hit_record* hit, mt19937* g) {
    const rt_scene_desc* sc = cx->sc;
    const rt_material* mat = &sc->materials[hit->material_id - 1];
    v3 color = V(0, 0, 0);
    v3 n_original = hit->n;
    v3 d_inc = ray->d;
    v3 n_shading = (dot(n_original, d_inc) < 0.0f) ? n_original : mul(n_original, -1);
    v3 x = hit->x;
    float eps = sc->shadow_ray_epsilon;
    v3 w0 = normalize(sub(ray->o, x));

    if (mat->type == RT_MAT_MIRROR) {
        v3 wr = perturb(normalize(sub(mul(mul(n_shading, 2), dot(n_shading, w0)), w0)), mat->roughness, g);
        v3 c = child_color(cx, ray, add(x, mul(n_shading, eps)), wr, g);
        color = add(color, emul(vld(mat->mirror_refl), c));
    } else if (mat->type == RT_MAT_CONDUCTOR) {
        v3 wr = perturb(normalize(sub(mul(mul(n_shading, 2), dot(n_shading, w0)), w0)), mat->roughness, g);
        float cosTheta = dot(w0, n_shading);
        float Fr = fresnel_conductor(cosTheta, mat->refraction_index, mat->absorption_index);
        v3 c = emul(vld(mat->mirror_refl), child_color(cx, ray, add(x, mul(n_shading, eps)), wr, g));
        color = add(color, mul(c, Fr));
    } else if (mat->type == RT_MAT_DIELECTRIC) {
        int entering = dot(n_original, ray->d) < 0.0f;
        v3 normal = entering ? n_original : mul(n_original, -1.0f);
        v3 w0l = normalize(sub(ray->o, x));
        float etaI = entering ? 1.0f : mat->refraction_index;
        float etaT = entering ? mat->refraction_index : 1.0f;
        float eta = etaI / etaT;
        float cosI = dot(w0l, normal);
        float one_minus = 1.0f - cosI * cosI;
        float sin2I = (0.0f < one_minus) ? one_minus : 0.0f; /* std::max(0.0f, v) */
        float sin2T = eta * eta * sin2I;
        if (sin2T >= 1.0f) {
            v3 wr = perturb(normalize(sub(mul(mul(normal, 2.0f), cosI), w0l)), mat->roughness, g);
            v3 c = child_color(cx, ray, add(x, mul(normal, eps)), wr, g);
            color = add(color, emul(vld(mat->mirror_refl), c));
        } else {
            float cosT = sqrtf(1.0f - sin2T);
            float Fr = fresnel_dielectric(cosI, cosT, etaI, etaT);
            v3 wr = perturb(normalize(sub(mul(mul(normal, 2.0f), cosI), w0l)), mat->roughness, g);
            v3 reflectColor = emul(vld(mat->mirror_refl), child_color(cx, ray, add(x, mul(normal, eps)), wr, g));
            v3 wt = perturb(normalize(add(mul(mul(w0l, -1.0f), eta), mul(normal, eta * cosI - cosT))),
                            mat->roughness, g);
            v3 refractColor = child_color(cx, ray, sub(x, mul(normal, eps)), wt, g);
            if (!entering) {
                float d = vlen(sub(x, ray->o));
                refractColor.x *= expf(-mat->absorption_coef[0] * d);
                refractColor.y *= expf(-mat->absorption_coef[1] * d);
                refractColor.z *= expf(-mat->absorption_coef[2] * d);
            }
            color = add(add(color, mul(reflectColor, Fr)), mul(refractColor, 1.0f - Fr));
        }
        if (!entering) return color;
    }

    color = add(color, emul(vld(sc->ambient_light), vld(mat->ambient_refl)));

    for (uint32_t l = 0; l < sc->n_point_lights; l++) {
        const rt_point_light* pl = &sc->point_lights[l];
        if (!in_shadow(cx, x, pl, n_shading, eps, ray->time)) color = add(color, diffuse_specular(mat, pl, x, n_shading, w0));
    }

    /* area lights, raytracer.cpp:1379-1433 */
    for (uint32_t l = 0; l < sc->n_area_lights; l++) {
        const rt_area_light* al = &sc->area_lights[l];
        float u = (mt_uniform(g) - 0.5f) * al->size;
        float v = (mt_uniform(g) - 0.5f) * al->size;
        v3 n_light = vld(al->normal), tangent, bitangent;
        orthonormal_basis(n_light, &tangent, &bitangent);
        v3 sample = add(add(vld(al->position), mul(tangent, u)), mul(bitangent, v));
        v3 to_light = sub(sample, x);
        float dist_to_light = vlen(to_light);
        v3 wi = divs(to_light, dist_to_light);
        ray_t sr;
        sr.o = add(x, mul(n_shading, eps));
        sr.d = wi;
        sr.depth = 0;
        sr.time = ray->time;
        hit_record sh;
        int in_sh = 0;
        if (find_closest_hit(cx, &sr, &sh, NULL)) {
            float dist_to_hit = vlen(sub(sh.x, sr.o));
            if (dist_to_hit < dist_to_light - eps) in_sh = 1;
        }
        if (!in_sh) {
            float cos_light = fabsf(dot(vld(al->normal), mul(wi, -1.0f)));
            float area = al->size * al->size;
            float solid_angle = (area * cos_light) / (dist_to_light * dist_to_light);
            v3 irradiance = mul(vld(al->radiance), solid_angle);
            float nd = dot(n_shading, wi);
            float cos_theta = (0.0f < nd) ? nd : 0.0f;
            v3 diffuse = mul(emul(vld(mat->diffuse_refl), irradiance), cos_theta);
            v3 h = normalize(add(wi, w0));
            float nh = dot(n_shading, h);
            float cos_alpha = (0.0f < nh) ? nh : 0.0f;
            float spec_factor = powf(cos_alpha, mat->phong_exponent);
            v3 specular = mul(emul(vld(mat->specular_refl), irradiance), spec_factor);
            color = add(add(color, diffuse), specular);
        }
    }
    return color;
}

/* ComputeColor, raytracer.cpp:382-401 */
static v3 compute_color(const ctx_t* cx, const ray_t* ray, mt19937* g) {
    if (ray->depth > cx->sc->max_recursion_depth) return V(0, 0, 0);
    if (ray->depth == 0) cx->ctr->rays_primary++;
    else cx->ctr->rays_secondary++;
    hit_record hit;
    if (find_closest_hit(cx, ray, &hit, NULL)) return apply_shading(cx, ray, &hit, g);
    if (ray->depth == 0) return vld(cx->sc->background_color);
    return V(0, 0, 0);
}

/* ComputeRay, raytracer.cpp:343-378 */
static ray_t compute_ray(const rt_camera_desc* cam, int j, int i, float jx, float jy, float au, float av, float time) {
    float su = (j + jx) * cam->pixel_width;
    float sv = (i + jy) * cam->pixel_height;
    v3 q = vld(cam->q), u = vld(cam->u), v = vld(cam->v), pos = vld(cam->position);
    v3 sample_point = sub(add(q, mul(u, su)), mul(v, sv));
    ray_t r;
    r.depth = 0;
    r.time = time;
    if (!cam->has_depth_of_field) {
        r.o = pos;
        r.d = normalize(sub(sample_point, pos));
    } else {
        v3 dir = normalize(sub(pos, sample_point));
        float t_focal = cam->focus_distance / dot(dir, mul(vld(cam->w), -1));
        v3 focal_point = add(pos, mul(dir, t_focal));
        float lens_u = (au - 0.5f) * cam->aperture_size;
        float lens_v = (av - 0.5f) * cam->aperture_size;
        v3 lens_sample = add(add(pos, mul(u, lens_u)), mul(v, lens_v));
        r.o = lens_sample;
        r.d = normalize(sub(focal_point, lens_sample));
    }
    return r;
}

static void ctr_add(ray_ctr* a, const ray_ctr* b) {
    a->rays += b->rays; a->top_nodes += b->top_nodes; a->mesh_nodes += b->mesh_nodes; a->tri_tests += b->tri_tests;
    a->sphere_tests += b->sphere_tests; a->plane_tests += b->plane_tests; a->inst_entries += b->inst_entries;
}

/* The pixel loop, raytracer.cpp:40-87, over the window [x0,x0+w) x [y0,y0+h).
 * rgb (nullable): row stride rgb_stride bytes.  hit_ids (nullable): 2 u32 per
 * pixel as documented for rt_render_tile in rt.h.  threads <= 0: OpenMP default. */
int rto_render(const rt_scene_desc* sc, const rt_camera_desc* cam, int x0, int y0, int w, int h, uint8_t* rgb,
               size_t rgb_stride, uint32_t* hit_ids, rto_counters* counters, int threads) {
    if (!sc || !cam || w < 0 || h < 0) return RT_ERR_INVALID;
    const int width = cam->image_width;
    const int N = cam->num_samples;
    const int sps = cam->samples_per_side;
    if (N < 1 || sps < 1) return RT_ERR_INVALID;
    const float inv_n = 1.0f / N;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    thread_ctr* tc = (thread_ctr*)calloc((size_t)nthreads, sizeof(thread_ctr));
    if (!tc) return RT_ERR_OOM;
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        ctx_t cx = {sc, cam, &tc[tid]};
        mt19937* g = (mt19937*)malloc(sizeof(mt19937));
        int* perm = (int*)malloc(sizeof(int) * (size_t)N);
#pragma omp for collapse(2) schedule(static)
        for (int i = y0; i < y0 + h; i++) {
            for (int j = x0; j < x0 + w; j++) {
                mt_seed(g, (uint32_t)(i * width + j));
                for (int k = 0; k < N; k++) perm[k] = k;
                shuffle_ints(perm, N, g);
                v3 sum = V(0.0f, 0.0f, 0.0f);
                for (int s = 0; s < N; s++) {
                    int sx = s % sps, sy = s / sps;
                    float jx = (sx + mt_uniform(g)) / sps;
                    float jy = (sy + mt_uniform(g)) / sps;
                    int ai = perm[s];
                    int ax = ai % sps, ay = ai / sps;
                    float au = (ax + mt_uniform(g)) / sps;
                    float av = (ay + mt_uniform(g)) / sps;
                    float time = mt_uniform(g);
                    ray_t ray = compute_ray(cam, j, i, jx, jy, au, av, time);
                    if (s == 0 && hit_ids) {
                        /* identity of the primary hit; counted separately so the
                         * render counters are not doubled */
                        thread_ctr scratch;
                        memset(&scratch, 0, sizeof(scratch));
                        ctx_t cx2 = {sc, cam, &scratch};
                        hit_record hr;
                        hit_info hi;
                        uint32_t* o = hit_ids + ((size_t)(i - y0) * (size_t)w + (size_t)(j - x0)) * 2;
                        if (find_closest_hit(&cx2, &ray, &hr, &hi)) {
                            o[0] = ((uint32_t)hi.kind << 28) | (uint32_t)hi.index;
                            o[1] = hi.kind == RT_KIND_MESH ? hi.face_idx : 0u;
                        } else {
                            o[0] = 0xFFFFFFFFu;
                            o[1] = 0xFFFFFFFFu;
                        }
                    }
                    if (rgb) sum = add(sum, compute_color(&cx, &ray, g));
                }
                if (rgb) {
                    v3 c = mul(sum, inv_n);
                    uint8_t* o = rgb + (size_t)(i - y0) * rgb_stride + (size_t)(j - x0) * 3;
                    o[0] = (unsigned char)clampf(c.x, 0.0f, 255.0f);
                    o[1] = (unsigned char)clampf(c.y, 0.0f, 255.0f);
                    o[2] = (unsigned char)clampf(c.z, 0.0f, 255.0f);
                }
            }
        }
        free(g);
        free(perm);
    }
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        ray_ctr c, s;
        memset(&c, 0, sizeof(c));
        memset(&s, 0, sizeof(s));
        for (int t = 0; t < nthreads; t++) {
            ctr_add(&c, &tc[t].c);
            ctr_add(&s, &tc[t].s);
            counters->rays_primary += tc[t].rays_primary;
            counters->rays_secondary += tc[t].rays_secondary;
            if (tc[t].max_stack > counters->max_stack) counters->max_stack = tc[t].max_stack;
        }
        counters->c_rays = c.rays; counters->c_top_nodes = c.top_nodes; counters->c_mesh_nodes = c.mesh_nodes;
        counters->c_tri_tests = c.tri_tests; counters->c_sphere_tests = c.sphere_tests;
        counters->c_plane_tests = c.plane_tests; counters->c_inst_entries = c.inst_entries;
        counters->s_rays = s.rays; counters->s_top_nodes = s.top_nodes; counters->s_mesh_nodes = s.mesh_nodes;
        counters->s_tri_tests = s.tri_tests; counters->s_sphere_tests = s.sphere_tests;
        counters->s_plane_tests = s.plane_tests; counters->s_inst_entries = s.inst_entries;
    }
    free(tc);
    return RT_OK;
}

/* Intermediates of one pixel's FIRST sample, for debugging a parity failure:
 * out[0..2] ray origin, [3..5] direction, [6] time, [7] hit?, [8] minT, [9] kind, [10] index,
 * [11..13] normal, [14..16] hit point, [17..19] n_shading, [20+l] !InShadow(light l) for l < 8. */
int rto_probe(const rt_scene_desc* sc, const rt_camera_desc* cam, int px, int py, float* out) {
    thread_ctr tc;
    memset(&tc, 0, sizeof(tc));
    ctx_t cx = {sc, cam, &tc};
    mt19937* g = (mt19937*)malloc(sizeof(mt19937));
    const int N = cam->num_samples, sps = cam->samples_per_side;
    int* perm = (int*)malloc(sizeof(int) * (size_t)N);
    mt_seed(g, (uint32_t)(py * cam->image_width + px));
    for (int k = 0; k < N; k++) perm[k] = k;
    shuffle_ints(perm, N, g);
    float jx = (0 + mt_uniform(g)) / sps, jy = (0 + mt_uniform(g)) / sps;
    int ai = perm[0];
    float au = (ai % sps + mt_uniform(g)) / sps, av = (ai / sps + mt_uniform(g)) / sps;
    float time = mt_uniform(g);
    ray_t ray = compute_ray(cam, px, py, jx, jy, au, av, time);
    memset(out, 0, 28 * sizeof(float));
    out[0] = ray.o.x; out[1] = ray.o.y; out[2] = ray.o.z; out[3] = ray.d.x; out[4] = ray.d.y; out[5] = ray.d.z; out[6] = time;
    hit_record hr;
    hit_info hi;
    if (find_closest_hit(&cx, &ray, &hr, &hi)) {
        out[7] = 1;
        out[9] = (float)hi.kind; out[10] = (float)hi.index;
        out[11] = hr.n.x; out[12] = hr.n.y; out[13] = hr.n.z;
        out[14] = hr.x.x; out[15] = hr.x.y; out[16] = hr.x.z;
        v3 ns = (dot(hr.n, ray.d) < 0.0f) ? hr.n : mul(hr.n, -1);
        out[17] = ns.x; out[18] = ns.y; out[19] = ns.z;
        for (uint32_t l = 0; l < sc->n_point_lights && l < 8; l++)
            out[20 + l] = in_shadow(&cx, hr.x, &sc->point_lights[l], ns, sc->shadow_ray_epsilon, time) ? 0.f : 1.f;
    }
    free(g);
    free(perm);
    return RT_OK;
}

/* RNG known-answer helper for tests: the first `n_u32` raw outputs of
 * mt19937(seed), then shuffle of [0,n_perm) with a fresh mt19937(seed), then
 * n_f uniform floats drawn right after that shuffle. */
int rto_rng_kat(uint32_t seed, uint32_t* raw, int n_u32, int* perm, int n_perm, float* f, int n_f) {
    mt19937* g = (mt19937*)malloc(sizeof(mt19937));
    if (!g) return RT_ERR_OOM;
    mt_seed(g, seed);
    for (int i = 0; i < n_u32; i++) raw[i] = mt_next(g);
    mt_seed(g, seed);
    for (int i = 0; i < n_perm; i++) perm[i] = i;
    shuffle_ints(perm, n_perm, g);
    for (int i = 0; i < n_f; i++) f[i] = mt_uniform(g);
    free(g);
    return RT_OK;
}
