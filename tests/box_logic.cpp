// Host build of raytracer_amd/csrc/rt_box.h (the device's ray/box and ray/primitive arithmetic, same text) against
// the oracle's restatements (oracle/rt_oracle.c: intersect_aabb, intersects_triangle, intersect_sphere,
// intersect_plane; raytracer.cpp:815-999) over adversarial operands.  Boxes: the verdict of box_test and of
// node_slabs4 (every slot, both plane orders) equals the reference's hit / miss, and both give the same entry
// distance.  Triangles, spheres, planes: same verdict, and on a hit the same bits of t (and beta, gamma).
// Exit code 0 = all equal.
// Built and run by tests/test_box_logic.py with clang++ -O1 -ffp-contract=off (ext_vector_type needs clang).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RT_BOX_HOST 1
#include "../raytracer_amd/csrc/rt_device.h"     // over tests/shim/hip/hip_runtime.h (-I tests/shim)
#include "../raytracer_amd/csrc/rt_box.h"

namespace orc {          // the oracle's statics (q_rsqrt, det3x3, ...) share names with the device's
extern "C" {
#include "../oracle/rt_oracle.c"
}
}
using orc::ray_t;
using orc::v3;

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main() {
    const float den = 1e-45f, sub = 1e-39f, tiny = 1.5e-38f, inf = RT_INF;
    const std::vector<float> dirs = {0.0f, -0.0f, den, -den, sub, -sub, tiny, -tiny, 1.0f, -1.0f, 0.3f, -0.7f, 1e30f, -1e30f};
    const std::vector<float> planes = {-2.0f, -1.0f, -0.0f, 0.0f, 0.5f, 1.0f, 3.0f};
    const std::vector<float> origins = {-5.0f, -2.0f, -1.0f, -0.5f, -0.0f, 0.0f, 0.25f, 0.5f, 1.0f, 2.0f, 3.0f, 7.0f, 1e-40f, -1e-40f};
    const std::vector<float> bounds = {3.402823466e+38F, 10.0f, 1.0f, 0.5f, 1e-3f, 0.0f, -1.0f, inf};
    unsigned long long cases = 0, bad = 0, odd_cases = 0;
    // one axis varies adversarially at a time, the two others take a few ordinary and a few degenerate settings
    struct Axis { float d, o, mn, mx; };
    std::vector<Axis> hard, easy;
    for (float d : dirs) for (float o : origins) for (float mn : planes) for (float mx : planes) if (mn <= mx) hard.push_back({d, o, mn, mx});
    easy = {{1.0f, -3.0f, -1.0f, 1.0f}, {-0.5f, 2.0f, -1.0f, 1.0f}, {0.0f, 0.0f, -1.0f, 1.0f}, {0.0f, 1.0f, -1.0f, 1.0f}, {-0.0f, 2.0f, -1.0f, 1.0f},
            {-sub, -1.0f, -1.0f, 1.0f}, {-sub, 0.0f, -1.0f, 1.0f}, {sub, 1.0f, -1.0f, 1.0f}, {0.25f, 0.0f, 0.0f, 0.0f}};
    for (int axis = 0; axis < 3; axis++)
        for (const Axis& h : hard)
            for (const Axis& e1 : easy)
                for (size_t e2i = 0; e2i < easy.size(); e2i += 2) {
                    const Axis& e2 = easy[e2i];
                    Axis ax[3];
                    ax[axis] = h; ax[(axis + 1) % 3] = e1; ax[(axis + 2) % 3] = e2;
                    for (float minT : bounds) {
                        ray_t r;
                        std::memset(&r, 0, sizeof(r));
                        r.o.x = ax[0].o; r.o.y = ax[1].o; r.o.z = ax[2].o;
                        r.d.x = ax[0].d; r.d.y = ax[1].d; r.d.z = ax[2].d;
                        const float bmin[3] = {ax[0].mn, ax[1].mn, ax[2].mn}, bmax[3] = {ax[0].mx, ax[1].mx, ax[2].mx};
                        const bool ref_hit = orc::intersect_aabb(&r, bmin, bmax, minT) != RAY_MISS;
                        ActiveRay ar;
                        ar.set(V3{r.o.x, r.o.y, r.o.z}, V3{r.d.x, r.d.y, r.d.z});
                        if (fminf(fminf(ar.ix, ar.iy), ar.iz) == -inf) odd_cases++;
                        float te = 0.f;
                        const bool h1 = box_test(ar, bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], minT, te);
                        // the record as the device reads it: near / far planes picked by the sign of 1/d; the box sits in slot k,
                        // the other slots hold another box
                        const bool sx = bits(ar.ix) >> 31, sy = bits(ar.iy) >> 31, sz = bits(ar.iz) >> 31;
                        for (int k = 0; k < 4; k++) {
                            float mn[3][4], mx[3][4];
                            for (int a = 0; a < 3; a++) for (int j = 0; j < 4; j++) { mn[a][j] = (j == k) ? bmin[a] : -9.0f + j; mx[a][j] = (j == k) ? bmax[a] : 9.0f + j; }
                            auto f4 = [](const float* p) { return float4{p[0], p[1], p[2], p[3]}; };
                            float te4[4];
                            bool hit4[4];
                            // as the node step calls them: the fast form unless the ray has an infinite 1/d
                            if (!ray_is_odd(ar))
                                node_slabs4(ar, f4(sx ? mx[0] : mn[0]), f4(sx ? mn[0] : mx[0]), f4(sy ? mx[1] : mn[1]), f4(sy ? mn[1] : mx[1]),
                                            f4(sz ? mx[2] : mn[2]), f4(sz ? mn[2] : mx[2]), minT, te4, hit4);
                            else
                                node_slabs4_any(ar, f4(sx ? mx[0] : mn[0]), f4(sx ? mn[0] : mx[0]), f4(sy ? mx[1] : mn[1]), f4(sy ? mn[1] : mx[1]),
                                                f4(sz ? mx[2] : mn[2]), f4(sz ? mn[2] : mx[2]), sx, sy, sz, minT, te4, hit4);
                            cases++;
                            const bool same_te = !h1 || bits(te4[k]) == bits(te) || (te4[k] == 0.0f && te == 0.0f);
                            if (hit4[k] != ref_hit || h1 != ref_hit || !same_te) {
                                if (bad++ < 10)
                                    std::printf("MISMATCH o=(%g %g %g) d=(%g %g %g) box=(%g %g %g)-(%g %g %g) minT=%g: reference %d box_test %d (te %g) node_slabs4[%d] %d (te %g)\n",
                                                r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], minT,
                                                (int)ref_hit, (int)h1, te, k, (int)hit4[k], te4[k]);
                            }
                        }
                    }
                }
    std::printf("boxes: %llu cases (%llu of them with 1/d = -inf on some axis), %llu mismatches\n", cases / 4, odd_cases, bad);

    // ---- triangles, spheres, planes: seeded random operands mixed with degenerate ones
    unsigned long long pcases = 0, pbad = 0, phits = 0;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 32); };
    auto uni = [&](float lo, float hi) { return lo + (hi - lo) * (float)(rnd() >> 8) * (1.0f / 16777216.0f); };
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-20f, -1e-20f, 1e-39f, 1e20f, 0.5f, 2.0f, 1e-8f, -1e-8f};
    auto val = [&](float lo, float hi) { return (rnd() % 5 == 0) ? special[rnd() % 12] : uni(lo, hi); };
    for (int it = 0; it < 3000000; it++) {
        ray_t r;
        std::memset(&r, 0, sizeof(r));
        r.o.x = val(-3, 3); r.o.y = val(-3, 3); r.o.z = val(-3, 3);
        r.d.x = val(-1, 1); r.d.y = val(-1, 1); r.d.z = val(-1, 1);
        const float minT = (rnd() % 4 == 0) ? bounds[rnd() % bounds.size()] : uni(0.0f, 8.0f);
        ActiveRay ar;
        ar.set(V3{r.o.x, r.o.y, r.o.z}, V3{r.d.x, r.d.y, r.d.z});
        {   // triangle: three vertices, sometimes coincident / collinear / through the origin of the ray
            rt_vertex vd[3];
            std::memset(vd, 0, sizeof(vd));
            for (int v = 0; v < 3; v++) for (int c = 0; c < 3; c++) vd[v].pos[c] = val(-3, 3);
            if (rnd() % 16 == 0) std::memcpy(vd[1].pos, vd[0].pos, 12);
            if (rnd() % 16 == 0) { vd[2].pos[0] = r.o.x; vd[2].pos[1] = r.o.y; vd[2].pos[2] = r.o.z; }
            rt_face f;
            std::memset(&f, 0, sizeof(f));
            f.i0 = 1; f.i1 = 2; f.i2 = 3;
            float rb = 0.f, rg = 0.f;
            const float rt_ = orc::intersect_triangle(&r, &f, vd, minT, &rb, &rg);
            const float* a = vd[0].pos; const float* b = vd[1].pos; const float* c = vd[2].pos;
            float t = 0.f, bb = 0.f, gg = 0.f;
            const bool h = tri_test(ar, a[0], a[1], a[2], a[0] - b[0], a[1] - b[1], a[2] - b[2], a[0] - c[0], a[1] - c[1], a[2] - c[2], minT, t, bb, gg);
            pcases++;
            const bool ref_hit = rt_ != RAY_MISS;
            phits += ref_hit;
            if (h != ref_hit || (h && (bits(t) != bits(rt_) || bits(bb) != bits(rb) || bits(gg) != bits(rg)))) {
                if (pbad++ < 10) std::printf("TRIANGLE MISMATCH reference %d (t %g) device %d (t %g) minT %g\n", (int)ref_hit, rt_, (int)h, t, minT);
            }
        }
        {   // sphere
            const v3 cen = {val(-3, 3), val(-3, 3), val(-3, 3)};
            const float rad = (rnd() % 8 == 0) ? special[rnd() % 12] : uni(0.01f, 2.0f);
            const float rt_ = orc::intersect_sphere(&r, cen, rad, minT);
            float t = 0.f;
            const bool h = sphere_test(ar, cen.x, cen.y, cen.z, rad, minT, t);
            pcases++;
            phits += rt_ != RAY_MISS;
            if (h != (rt_ != RAY_MISS) || (h && bits(t) != bits(rt_))) {
                if (pbad++ < 10) std::printf("SPHERE MISMATCH reference t %g device %d (t %g)\n", rt_, (int)h, t);
            }
        }
        {   // plane
            DPlane p;
            std::memset(&p, 0, sizeof(p));
            p.nx = val(-1, 1); p.ny = val(-1, 1); p.nz = val(-1, 1); p.d = val(-3, 3);
            const v3 n = {p.nx, p.ny, p.nz};
            const float rt_ = orc::intersect_plane(&r, n, p.d, minT);
            float t = 0.f;
            const bool h = plane_test(V3{r.o.x, r.o.y, r.o.z}, V3{r.d.x, r.d.y, r.d.z}, p, minT, t);
            pcases++;
            phits += rt_ != RAY_MISS;
            if (h != (rt_ != RAY_MISS) || (h && bits(t) != bits(rt_))) {
                if (pbad++ < 10) std::printf("PLANE MISMATCH reference t %g device %d (t %g)\n", rt_, (int)h, t);
            }
        }
    }
    std::printf("triangles / spheres / planes: %llu cases (%llu hits), %llu mismatches\n", pcases, phits, pbad);
    bad += pbad;

    // ---- the per-object ray set-up and the way back to a world distance (raytracer.cpp:559-606): object_space_ray,
    // world_t against the oracle's, over random affine transforms (non-uniform and negative scales, shears, large
    // translations), with and without motion blur, on rays of unit and non-unit direction
    unsigned long long xbad = 0, xcases = 0;
    {
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) & 0xFFFFFF) / 16777216.0f; };
        auto sym = [&](float s) { return (rnd() * 2.0f - 1.0f) * s; };
        for (int k = 0; k < 400000; k++) {
            DObject ob;
            std::memset(&ob, 0, sizeof(ob));
            const float big = (k % 5 == 0) ? 1000.0f : 3.0f;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) {
                ob.M[r * 4 + c] = (c == 3) ? sym(big) : sym(2.0f) + ((r == c) ? ((k % 3) ? 1.0f : -1.0f) : 0.0f);
                ob.Minv[r * 4 + c] = (c == 3) ? sym(big) : sym(2.0f) + ((r == c) ? 1.0f : 0.0f);
            }
            ob.M[15] = ob.Minv[15] = 1.0f;
            ob.mbx = sym(2.0f); ob.mby = sym(2.0f); ob.mbz = sym(2.0f);
            ob.flags = (uint32_t)(k & 3);             // RT_OBJ_HAS_TRANSFORM = 1, RT_OBJ_HAS_MOTION_BLUR = 2
            ray_t ray;
            ray.o = v3{sym(20.0f), sym(20.0f), sym(20.0f)};
            v3 d = {sym(1.0f), sym(1.0f), sym(1.0f)};
            if (k % 2) { const float l = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z) + 1e-9f; d = v3{d.x / l, d.y / l, d.z / l}; }
            if (k % 17 == 0) d = v3{0.0f, (k % 34) ? 1.0f : -1.0f, 0.0f};
            ray.d = d;
            ray.depth = 0;
            ray.time = rnd();
            const float minT = (k % 11 == 0) ? FLT_MAX : rnd() * 100.0f;
            const float mb[3] = {ob.mbx, ob.mby, ob.mbz};
            ray_t test;
            float o_minT;
            orc::ray_ctr rc;
            std::memset(&rc, 0, sizeof(rc));
            orc::object_space_ray(&ray, ob.flags, ob.Minv, mb, minT, &test, &o_minT, &rc);
            V3 to, td;
            float d_minT;
            object_space_ray(ob, V3{ray.o.x, ray.o.y, ray.o.z}, V3{ray.d.x, ray.d.y, ray.d.z}, ray.time, minT, to, td, d_minT);
            const float t = rnd() * 50.0f;
            const float o_wt = orc::world_t(&ray, &test, t, ob.flags, ob.M, mb);
            const float d_wt = world_t(ob, V3{ray.o.x, ray.o.y, ray.o.z}, ray.time, to, td, t);
            xcases++;
            const bool same = bits(to.x) == bits(test.o.x) && bits(to.y) == bits(test.o.y) && bits(to.z) == bits(test.o.z) &&
                              bits(td.x) == bits(test.d.x) && bits(td.y) == bits(test.d.y) && bits(td.z) == bits(test.d.z) &&
                              bits(d_minT) == bits(o_minT) && bits(d_wt) == bits(o_wt);
            if (!same && xbad++ < 10) std::printf("TRANSFORM MISMATCH case %d flags %u: minT %g / %g, world t %g / %g\n", k, ob.flags, d_minT, o_minT, d_wt, o_wt);
        }
    }
    std::printf("object-space rays / world distances: %llu cases, %llu mismatches\n", xcases, xbad);
    bad += xbad;
    std::printf("total: %llu mismatches\n", bad);
    return bad ? 1 : 0;
}
