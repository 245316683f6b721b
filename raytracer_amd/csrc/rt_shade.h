// rt_shade.h — the arithmetic of ComputeRay, ApplyShading's set-up, the two Fresnel terms, ComputeDiffuseAndSpecular,
// CreateOrthonormalBasis and PerturbReflection (raytracer.cpp:343-378, 1250-1345, 1438-1537) in the reference's
// expression order.  No wave-level code and no device memory: tests/shade_logic.cpp compiles this very text for the host
// (over tests/shim/hip/hip_runtime.h) and compares it, bit for bit, with the oracle's restatement of the reference on
// random and adversarial operands.  Device builds include it through rt_kernels.h, after rt_device.h and rt_mt.h.
#pragma once

// ComputeRay, raytracer.cpp:343-378
__device__ __forceinline__ void compute_ray(const DCamera& cam, int j, int i, float jx, float jy, float au, float av,
                                            V3& o, V3& d) {
    float su = (j + jx) * cam.pixel_w;
    float sv = (i + jy) * cam.pixel_h;
    V3 q = ldv(cam.q), u = ldv(cam.u), v = ldv(cam.v), pos = ldv(cam.pos);
    V3 sample_point = (q + u * su) - v * sv;
    if (!cam.dof) {
        o = pos;
        d = normalize(sample_point - pos);
    } else {
        V3 dir = normalize(pos - sample_point);
        float t_focal = cam.focus / dot(dir, ldv(cam.w) * -1.0f);
        V3 focal_point = pos + dir * t_focal;
        float lens_u = (au - 0.5f) * cam.aperture;
        float lens_v = (av - 0.5f) * cam.aperture;
        V3 lens_sample = (pos + u * lens_u) + v * lens_v;
        o = lens_sample;
        d = normalize(focal_point - lens_sample);
    }
}

// Everything ApplyShading derives from (ray, hit point, normal, material) before
// it needs child colours or shadow results (raytracer.cpp:1250-1345).
struct ShadeCtx {
    V3 x, n_shading, w0;
    uint32_t type;
    // mirror / conductor
    V3 wr;                 // reflection direction
    V3 refl_origin;
    float Fr;              // conductor / dielectric Fresnel
    // dielectric
    bool entering, tir;
    V3 wt, refr_origin;
};

// Fresnel_Dielectric, raytracer.cpp:1459-1470
__device__ __forceinline__ float fresnel_dielectric(float cosTheta, float cosPhi, float n1, float n2) {
    float r_par = (n2 * cosTheta - n1 * cosPhi) / (n2 * cosTheta + n1 * cosPhi);
    float r_perp = (n1 * cosTheta - n2 * cosPhi) / (n1 * cosTheta + n2 * cosPhi);
    return (r_par * r_par + r_perp * r_perp) * 0.5f;
}
// Fresnel_Conductor, raytracer.cpp:1472-1488
__device__ __forceinline__ float fresnel_conductor(float cosTheta, float n, float k) {
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

__device__ __forceinline__ void shade_ctx(const DScene& S, const DMaterial& mat, V3 o, V3 d, float tHit, V3 n_original,
                                          ShadeCtx& c) {
    c.x = o + d * tHit;
    c.n_shading = (dot(n_original, d) < 0.0f) ? n_original : n_original * -1.0f;
    c.w0 = normalize(o - c.x);
    c.type = mat.type;
    c.entering = true;
    c.tir = false;
    c.Fr = 0.f;
    const float eps = S.shadow_eps;
    if (mat.type == RT_MAT_MIRROR || mat.type == RT_MAT_CONDUCTOR) {
        c.wr = normalize((c.n_shading * 2.0f) * dot(c.n_shading, c.w0) - c.w0);   // :1274 / :1286
        c.refl_origin = c.x + c.n_shading * eps;
        if (mat.type == RT_MAT_CONDUCTOR) c.Fr = fresnel_conductor(dot(c.w0, c.n_shading), mat.ior, mat.absorb_index);
    } else if (mat.type == RT_MAT_DIELECTRIC) {                                    // :1301-1362
        c.entering = dot(n_original, d) < 0.0f;
        V3 normal = c.entering ? n_original : n_original * -1.0f;
        float etaI = c.entering ? 1.0f : mat.ior;
        float etaT = c.entering ? mat.ior : 1.0f;
        float eta = etaI / etaT;
        float cosI = dot(c.w0, normal);
        float one_minus = 1.0f - cosI * cosI;
        float sin2I = (0.0f < one_minus) ? one_minus : 0.0f;
        float sin2T = eta * eta * sin2I;
        c.wr = normalize((normal * 2.0f) * cosI - c.w0);
        c.refl_origin = c.x + normal * eps;
        if (sin2T >= 1.0f) {
            c.tir = true;
        } else {
            float cosT = sqrtf(1.0f - sin2T);
            c.Fr = fresnel_dielectric(cosI, cosT, etaI, etaT);
            c.wt = normalize((c.w0 * -1.0f) * eta + normal * (eta * cosI - cosT));
            c.refr_origin = c.x - normal * eps;
        }
    }
}

// ComputeDiffuseAndSpecular, raytracer.cpp:1490-1520
__device__ __forceinline__ V3 diffuse_specular(const DMaterial& mat, const DPointLight& L, V3 point, V3 normal, V3 w0) {
    V3 Lv = mk(L.px, L.py, L.pz) - point;
    V3 wi = normalize(Lv);
    float cos_theta = dot(wi, normal);
    if (cos_theta < 0.0f) return mk(0.0f, 0.0f, 0.f);
    float d_sq = dot(Lv, Lv);
    V3 irradiance = vdiv(mk(L.ix, L.iy, L.iz), d_sq);
    V3 diffuse = emul(ldv(mat.kd) * cos_theta, irradiance);
    V3 h = normalize(wi + w0);
    float cos_alpha = dot(normal, h);
    if (cos_alpha < 0.0f) return diffuse;
    V3 spec = emul(ldv(mat.ks) * rt_powf(cos_alpha, mat.phong), irradiance);
    return diffuse + spec;
}

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
