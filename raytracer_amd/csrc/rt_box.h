// rt_box.h — the ray/box and ray/primitive arithmetic of the traversal (IntersectAABB, IntersectsTriangle_Bary,
// IntersectSphere, IntersectsPlane: raytracer.cpp:815-999; the per-object ray set-up and the way back to a world
// distance: raytracer.cpp:559-606), kept free of wave-level constructs so that
// tests/box_logic.cpp can compile the very same text for the host and compare it with the oracle's restatement
// of the reference over adversarial operands (zero and denormal direction components, origins on box planes,
// degenerate boxes and triangles, infinities).  Device builds include it through rt_traverse.h, after
// rt_device.h; the host test defines RT_BOX_HOST and includes rt_device.h over tests/shim/hip/hip_runtime.h.
#pragma once
#ifdef RT_BOX_HOST
#define RT_BOX_FN static inline
#define RT_BOX_MEMFN inline
#define RT_BOX_ANY(x) (x)
#else
#define RT_BOX_FN __device__ __forceinline__
#define RT_BOX_MEMFN __device__ __forceinline__
#define RT_BOX_ANY(x) __any(x)
#endif

// The ray a lane is currently traversing with (world or object space).
struct ActiveRay {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;     // 1.0f / d (IEEE); +inf where d == 0 (such rays, and those whose 1/d overflows, take box_literal)
    RT_BOX_MEMFN void set(V3 o, V3 d) {
        ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
        ix = (d.x == 0.0f) ? RT_INF : 1.0f / d.x;
        iy = (d.y == 0.0f) ? RT_INF : 1.0f / d.y;
        iz = (d.z == 0.0f) ? RT_INF : 1.0f / d.z;
    }
};

// IntersectAABB (raytracer.cpp:815-894), literally: the path of rays with an infinite 1/d on some axis (d == 0,
// or a d so small that 1/d overflows).  The reference treats d == 0 by an inside test that does not depend on
// minT, and 1/d = +-inf produces NaN products on box planes through the origin, whose fate depends on the
// unswapped order — neither is reproduced by the ordered-slab arithmetic below once minT itself is infinite
// (testMinT = minT * scale overflows for a scaled instance, raytracer.cpp:579) or 1/d is -inf.  The early returns
// become a flag: later axes only touch values that are no longer looked at.  tEnter = the reference's tMin.
RT_BOX_FN bool box_literal(const ActiveRay& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float minT,
                           float& tEnter) {
    float tMin = 0.0f, tMax = minT;
    bool miss = false;
    auto axis = [&](float d, float inv, float o, float mn, float mx) {
        if (d != 0.0f) {
            float t0 = (mn - o) * inv, t1 = (mx - o) * inv;
            if (t1 < t0) { const float t = t0; t0 = t1; t1 = t; }
            tMin = (tMin < t0) ? t0 : tMin;     // std::max(tMin, t0)
            tMax = (t1 < tMax) ? t1 : tMax;     // std::min(tMax, t1)
            if (tMax < tMin) miss = true;
        } else if (o < mn || o > mx) {
            miss = true;
        }
    };
    axis(r.dx, r.ix, r.ox, mnx, mxx);
    axis(r.dy, r.iy, r.oy, mny, mxy);
    axis(r.dz, r.iz, r.oz, mnz, mxz);
    tEnter = tMin;
    if (miss) return false;
    float tHit = tMin;
    if (!(tMin > 0.0f)) {
        if (tMax <= 0.0f) return false;
        tHit = tMax;
    }
    return !(tHit > minT);
}
RT_BOX_FN bool ray_is_odd(const ActiveRay& r) {     // some 1/d is infinite
    return fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz)) == RT_INF;
}

// The same verdict + entry distance, branch-free, for rays whose three 1/d are finite: no product is a NaN
// then, so per axis lo = min(t0,t1), hi = max(t0,t1) as the reference's swap leaves them; tMin = max(0, lo...),
// tMax = min(minT, hi...); the per-axis early returns equal one final test because tMin only grows and tMax
// only shrinks; "hit" = !(tMax < tMin) && !(tMin == 0 && tMax <= 0) = (tMin <= tMax) && (tMax > 0); the
// trailing `tHit > minT` test (raytracer.cpp:889) cannot fire once tMax <= minT.
RT_BOX_FN bool box_test(const ActiveRay& r, float mnx, float mny, float mnz, float mxx, float mxy,
                                         float mxz, float tmax, float& tEnter) {
    if (ray_is_odd(r)) return box_literal(r, mnx, mny, mnz, mxx, mxy, mxz, tmax, tEnter);
    const float ax = (mnx - r.ox) * r.ix, bx = (mxx - r.ox) * r.ix;
    const float ay = (mny - r.oy) * r.iy, by = (mxy - r.oy) * r.iy;
    const float az = (mnz - r.oz) * r.iz, bz = (mxz - r.oz) * r.iz;
    const bool sx = bx < ax, sy = by < ay, sz = bz < az;
    const float lox = sx ? bx : ax, hix = sx ? ax : bx;
    const float loy = sy ? by : ay, hiy = sy ? ay : by;
    const float loz = sz ? bz : az, hiz = sz ? az : bz;
    const float tMin = fmaxf(fmaxf(fmaxf(0.0f, lox), loy), loz);
    const float tMax = fminf(fminf(fminf(tmax, hix), hiy), hiz);
    tEnter = tMin;
    return (tMin <= tMax) && (tMax > 0.0f);
}

// The same test for the four slots of a node record at once.  n* / f* hold, per axis, the planes of the four
// slots the ray meets first / last when they are picked by the sign of 1/d (min then max for 1/d >= 0): with
// a = (near - o) * inv and b = (far - o) * inv the reference's t0,t1 are (a,b) or (b,a), and since FP32
// subtraction and multiplication by a finite inv are monotone, a <= b: its swap leaves lo = a, hi = b (equal
// values may stay unswapped there, which changes nothing).  Lanes whose ray has an infinite 1/d take
// box_literal on the same planes instead.  The two products per axis are packed FP32 operations on slot pairs
// (IEEE per component); min/max chains as in box_test.
typedef float v2f __attribute__((ext_vector_type(2)));
RT_BOX_FN void slab_pair(float p0, float p1, float o, float inv, float& r0, float& r1) {
    const v2f oo = {o, o}, ii = {inv, inv}, pp = {p0, p1};
    const v2f r = (pp - oo) * ii;
    r0 = r.x;
    r1 = r.y;
}
// (fast form: the caller has made sure that no lane taking part has an infinite 1/d)
RT_BOX_FN void node_slabs4(const ActiveRay& r, float4 nx, float4 fx, float4 ny, float4 fy, float4 nz, float4 fz,
                           float tmax, float te[4], bool hit[4]) {
    float ax[4], bx[4], ay[4], by[4], az[4], bz[4];
    slab_pair(nx.x, nx.y, r.ox, r.ix, ax[0], ax[1]); slab_pair(nx.z, nx.w, r.ox, r.ix, ax[2], ax[3]);
    slab_pair(fx.x, fx.y, r.ox, r.ix, bx[0], bx[1]); slab_pair(fx.z, fx.w, r.ox, r.ix, bx[2], bx[3]);
    slab_pair(ny.x, ny.y, r.oy, r.iy, ay[0], ay[1]); slab_pair(ny.z, ny.w, r.oy, r.iy, ay[2], ay[3]);
    slab_pair(fy.x, fy.y, r.oy, r.iy, by[0], by[1]); slab_pair(fy.z, fy.w, r.oy, r.iy, by[2], by[3]);
    slab_pair(nz.x, nz.y, r.oz, r.iz, az[0], az[1]); slab_pair(nz.z, nz.w, r.oz, r.iz, az[2], az[3]);
    slab_pair(fz.x, fz.y, r.oz, r.iz, bz[0], bz[1]); slab_pair(fz.z, fz.w, r.oz, r.iz, bz[2], bz[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float tMin = fmaxf(fmaxf(fmaxf(0.0f, ax[k]), ay[k]), az[k]);
        const float tMax = fminf(fminf(fminf(tmax, bx[k]), by[k]), bz[k]);
        te[k] = tMin;
        hit[k] = (tMin <= tMax) && (tMax > 0.0f);
    }
}
// (general form: lanes with an infinite 1/d take box_literal on the same planes; sx/sy/sz = the plane order picked)
RT_BOX_FN void node_slabs4_any(const ActiveRay& r, float4 nx, float4 fx, float4 ny, float4 fy, float4 nz, float4 fz,
                               bool sx, bool sy, bool sz, float tmax, float te[4], bool hit[4]) {
    if (ray_is_odd(r)) {
        const float nxa[4] = {nx.x, nx.y, nx.z, nx.w}, fxa[4] = {fx.x, fx.y, fx.z, fx.w};
        const float nya[4] = {ny.x, ny.y, ny.z, ny.w}, fya[4] = {fy.x, fy.y, fy.z, fy.w};
        const float nza[4] = {nz.x, nz.y, nz.z, nz.w}, fza[4] = {fz.x, fz.y, fz.z, fz.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            hit[k] = box_literal(r, sx ? fxa[k] : nxa[k], sy ? fya[k] : nya[k], sz ? fza[k] : nza[k],
                                 sx ? nxa[k] : fxa[k], sy ? nya[k] : fya[k], sz ? nza[k] : fza[k], tmax, te[k]);
    } else {
        node_slabs4(r, nx, fx, ny, fy, nz, fz, tmax, te, hit);
    }
}

// IntersectsTriangle_Bary, raytracer.cpp:896-947, with col_A0/col_A1 preloaded.
// Straight-line form: the reference's early returns become one verdict (each
// condition is evaluated on the same values in the same order; a quotient by a
// detA that failed the first test is computed but never looked at), so a wave
// whose lanes test different triangles executes no divergent branches.
RT_BOX_FN bool tri_test(const ActiveRay& r, float ax, float ay, float az, float e1x, float e1y,
                                         float e1z, float e2x, float e2y, float e2z, float minT, float& t_out,
                                         float& beta_out, float& gamma_out) {
    const float bx = ax - r.ox, by = ay - r.oy, bz = az - r.oz;
    const float detA = det3x3(e1x, e2x, r.dx, e1y, e2y, r.dy, e1z, e2z, r.dz);
    const float beta = det3x3(bx, e2x, r.dx, by, e2y, r.dy, bz, e2z, r.dz) / detA;
    const float gamma = det3x3(e1x, bx, r.dx, e1y, by, r.dy, e1z, bz, r.dz) / detA;
    const float t = det3x3(e1x, e2x, bx, e1y, e2y, by, e1z, e2z, bz) / detA;
    const bool rej = (fabsf(detA) < RT_EPS_PARALLEL) | (beta < -RT_EPS_PARALLEL) | (gamma < -RT_EPS_PARALLEL) |
                     ((beta + gamma) > (1.0f + RT_EPS_PARALLEL)) | (t < -RT_EPS_PARALLEL) | (t >= minT);
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return !rej;
}

// IntersectSphere, raytracer.cpp:949-980 (sqrt(double(delta)) narrowed to float
// is the correctly rounded sqrtf: 53 >= 2*24+2).
RT_BOX_FN bool sphere_test(const ActiveRay& r, float cx, float cy, float cz, float radius, float minT,
                                            float& t_out) {
    V3 oc = mk(r.ox - cx, r.oy - cy, r.oz - cz);
    V3 d = mk(r.dx, r.dy, r.dz);
    float A = dot(d, d);
    float B = 2 * dot(d, oc);
    float C = dot(oc, oc) - radius * radius;
    float delta = B * B - 4 * A * C;
    if (delta < 0.0f) return false;
    float sq = sqrtf(delta);
    float t1 = (-B - sq) / (2 * A);
    float t2 = (-B + sq) / (2 * A);
    float t;
    if (t1 > 0.0f) t = t1;
    else if (t2 > 0.0f) t = t2;
    else return false;
    if (t >= minT) return false;
    t_out = t;
    return true;
}

// IntersectsPlane, raytracer.cpp:982-999
RT_BOX_FN bool plane_test(V3 o, V3 d, const DPlane& p, float minT, float& t_out) {
    V3 n = mk(p.nx, p.ny, p.nz);
    float den = dot(d, n);
    if (fabsf(den) < RT_EPS_PARALLEL) return false;
    float t = -(dot(o, n) + p.d) / den;
    if (t < 0.0f || t >= minT) return false;
    t_out = t;
    return true;
}

// Per-object ray set-up, raytracer.cpp:559-587 (and :1070-1095 for shadows).
RT_BOX_FN void object_space_ray(const DObject& ob, V3 ro, V3 rd, float time, float minT, V3& to,
                                                 V3& td, float& testMinT) {
    to = ro;
    td = rd;
    testMinT = minT;
    if (ob.flags & RT_OBJ_HAS_TRANSFORM) {
        V3 o2 = ro;
        if (ob.flags & RT_OBJ_HAS_MOTION_BLUR) o2 = o2 - mk(ob.mbx, ob.mby, ob.mbz) * time;
        to = xf_point(ob.Minv, o2);
        V3 objDir = xf_vector(ob.Minv, rd);
        td = normalize(objDir);
        float scale = vlen(objDir);
        testMinT = minT * scale;
    } else if (ob.flags & RT_OBJ_HAS_MOTION_BLUR) {
        to = ro - mk(ob.mbx, ob.mby, ob.mbz) * time;
    }
}
// raytracer.cpp:598-606
RT_BOX_FN float world_t(const DObject& ob, V3 ro, float time, V3 to, V3 td, float t) {
    if (!(ob.flags & RT_OBJ_HAS_TRANSFORM)) return t;
    V3 objHit = to + td * t;
    V3 worldHit = xf_point(ob.M, objHit);
    if (ob.flags & RT_OBJ_HAS_MOTION_BLUR) worldHit = worldHit + mk(ob.mbx, ob.mby, ob.mbz) * time;
    return vlen(worldHit - ro);
}
