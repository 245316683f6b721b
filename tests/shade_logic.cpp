// Host build of raytracer_amd/csrc/rt_shade.h (the device's ray generation and shading arithmetic, same text) against
// the oracle's restatement of the reference (oracle/rt_oracle.c: compute_ray, fresnel_dielectric, fresnel_conductor,
// diffuse_specular, orthonormal_basis, perturb; raytracer.cpp:343-378, 1438-1537): same bits on random operands and on
// the edges (grazing and back-facing normals, zero roughness, axis-aligned directions, lights inside the surface,
// depth of field on and off).  Exit code 0 = all equal.  Built and run by tests/test_shade_logic.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
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
#include "../raytracer_amd/csrc/rt_mt.h"
#include "../raytracer_amd/csrc/rt_shade.h"

namespace orc {
extern "C" {
#include "../oracle/rt_oracle.c"
}
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static long mismatches = 0, checked = 0;
static void eq(float a, float b, const char* what, long k) {
    checked++;
    if (bits(a) != bits(b) && !(std::isnan(a) && std::isnan(b))) {
        if (mismatches < 20) printf("%s case %ld: device %.9g (%08x) oracle %.9g (%08x)\n", what, k, a, bits(a), b, bits(b));
        mismatches++;
    }
}
static void eq3(V3 a, orc::v3 b, const char* what, long k) { eq(a.x, b.x, what, k); eq(a.y, b.y, what, k); eq(a.z, b.z, what, k); }

int main() {
    std::mt19937 rng(20240611);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f), P(0.0f, 1.0f);
    auto rv = [&](float s) { return mk(U(rng) * s, U(rng) * s, U(rng) * s); };
    auto unit = [&]() { V3 v; float l; do { v = rv(1.0f); l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); } while (l < 1e-3f); return mk(v.x / l, v.y / l, v.z / l); };
    const V3 axes[] = {mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1), mk(-1, 0, 0), mk(0, -1, 0), mk(0, 0, -1), mk(0.70710678f, 0.70710678f, 0), mk(0, 0.70710678f, -0.70710678f),
                       mk(0.57735027f, 0.57735027f, 0.57735027f), mk(1e-20f, 1.0f, 1e-20f), mk(1.0f, 1e-8f, -1e-8f)};

    // ---- ComputeRay: pinhole and thin lens
    for (long k = 0; k < 200000; k++) {
        rt_camera_desc c;
        std::memset(&c, 0, sizeof(c));
        V3 pos = rv(50.0f), u = unit(), v = unit(), w = unit(), q = rv(60.0f);
        const float* src[5] = {&pos.x, &u.x, &v.x, &w.x, &q.x};
        float* dst[5] = {c.position, c.u, c.v, c.w, c.q};
        for (int a = 0; a < 5; a++) for (int b = 0; b < 3; b++) dst[a][b] = src[a][b];
        c.pixel_width = P(rng) * 0.01f + 1e-4f; c.pixel_height = P(rng) * 0.01f + 1e-4f;
        c.image_width = 1920; c.image_height = 1080; c.num_samples = 16; c.samples_per_side = 4;
        c.aperture_size = P(rng) * 0.5f; c.focus_distance = P(rng) * 20.0f + 0.1f;
        c.has_depth_of_field = (k & 1);
        DCamera d;
        std::memcpy(d.pos, c.position, 12); std::memcpy(d.u, c.u, 12); std::memcpy(d.v, c.v, 12); std::memcpy(d.w, c.w, 12); std::memcpy(d.q, c.q, 12);
        d.pixel_w = c.pixel_width; d.pixel_h = c.pixel_height; d.image_w = c.image_width; d.image_h = c.image_height;
        d.num_samples = c.num_samples; d.sps = c.samples_per_side; d.aperture = c.aperture_size; d.focus = c.focus_distance; d.dof = c.has_depth_of_field;
        const int j = (int)(P(rng) * 1920), i = (int)(P(rng) * 1080);
        const float jx = P(rng), jy = P(rng), au = P(rng), av = P(rng);
        V3 o, dir;
        compute_ray(d, j, i, jx, jy, au, av, o, dir);
        const orc::ray_t r = orc::compute_ray(&c, j, i, jx, jy, au, av, 0.5f);
        eq3(o, r.o, "compute_ray origin", k);
        eq3(dir, r.d, "compute_ray direction", k);
    }

    // ---- the Fresnel terms
    const float cosv[] = {0.0f, 1e-30f, 1e-6f, 0.1f, 0.5f, 0.70710678f, 0.99999994f, 1.0f, -0.3f};
    const float iors[] = {0.0f, 0.17f, 0.37f, 1.0f, 1.33f, 1.5f, 2.4f, 10.0f};
    long fk = 0;
    for (float ct : cosv) for (float cp : cosv) for (float n1 : iors) for (float n2 : iors) {
        eq(fresnel_dielectric(ct, cp, n1, n2), orc::fresnel_dielectric(ct, cp, n1, n2), "fresnel_dielectric", fk);
        eq(fresnel_conductor(ct, n1, n2), orc::fresnel_conductor(ct, n1, n2), "fresnel_conductor", fk++);
    }
    for (long k = 0; k < 300000; k++) {
        const float ct = P(rng), cp = P(rng), n1 = P(rng) * 3.0f, n2 = P(rng) * 3.0f;
        eq(fresnel_dielectric(ct, cp, n1, n2), orc::fresnel_dielectric(ct, cp, n1, n2), "fresnel_dielectric (random)", k);
        eq(fresnel_conductor(ct, n1, n2 * 2.0f), orc::fresnel_conductor(ct, n1, n2 * 2.0f), "fresnel_conductor (random)", k);
    }

    // ---- ComputeDiffuseAndSpecular (powf inside: glibc's on the oracle's side, its restatement on the device's)
    for (long k = 0; k < 300000; k++) {
        rt_material m;
        std::memset(&m, 0, sizeof(m));
        DMaterial dm;
        std::memset(&dm, 0, sizeof(dm));
        for (int a = 0; a < 3; a++) { m.diffuse_refl[a] = dm.kd[a] = P(rng); m.specular_refl[a] = dm.ks[a] = P(rng); }
        const float phs[] = {0.0f, 1.0f, 2.0f, 3.5f, 10.0f, 40.0f, 80.0f, 100.0f, 1000.0f};
        m.phong_exponent = dm.phong = (k % 3 == 0) ? phs[(k / 3) % 9] : P(rng) * 200.0f;
        rt_point_light L;
        DPointLight dl;
        V3 lp = rv(20.0f), li = mk(P(rng) * 9000.0f, P(rng) * 9000.0f, P(rng) * 9000.0f);
        L.position[0] = dl.px = lp.x; L.position[1] = dl.py = lp.y; L.position[2] = dl.pz = lp.z;
        L.intensity[0] = dl.ix = li.x; L.intensity[1] = dl.iy = li.y; L.intensity[2] = dl.iz = li.z;
        V3 x = (k % 50 == 0) ? lp : rv(20.0f);                  // (a light exactly at the shaded point: 0 / 0)
        V3 n = (k % 7 == 0) ? axes[(k / 7) % 11] : unit(), w0 = (k % 11 == 0) ? axes[(k / 11) % 11] : unit();
        const V3 a = diffuse_specular(dm, dl, x, n, w0);
        const orc::v3 b = orc::diffuse_specular(&m, &L, orc::v3{x.x, x.y, x.z}, orc::v3{n.x, n.y, n.z}, orc::v3{w0.x, w0.y, w0.z});
        eq3(a, b, "diffuse_specular", k);
    }

    // ---- CreateOrthonormalBasis and PerturbReflection (two draws from the pixel's generator each)
    std::vector<uint32_t> mem(624);
    for (long k = 0; k < 200000; k++) {
        const V3 r = (k % 5 == 0) ? axes[(k / 5) % 11] : unit();
        V3 t, b;
        orthonormal_basis(r, t, b);
        orc::v3 ot, ob;
        orc::orthonormal_basis(orc::v3{r.x, r.y, r.z}, &ot, &ob);
        eq3(t, ot, "orthonormal_basis tangent", k);
        eq3(b, ob, "orthonormal_basis bitangent", k);
        const uint32_t seed = (uint32_t)rng();
        MtGlobal g;
        g.mt = mem.data(); g.stride = 1;
        g.seed(seed);
        orc::mt19937 og;
        orc::mt_seed(&og, seed);
        const float rough = (k % 9 == 0) ? 0.0f : P(rng) * 0.5f;
        for (int rep = 0; rep < 3; rep++) {
            const V3 pr = perturb_reflection(r, rough, g);
            const orc::v3 po = orc::perturb(orc::v3{r.x, r.y, r.z}, rough, &og);
            eq3(pr, po, "perturb_reflection", k);
        }
    }
    printf("checked %ld values; total: %ld mismatches\n", checked, mismatches);
    return mismatches ? 1 : 0;
}
