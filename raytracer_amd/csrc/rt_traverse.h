// rt_traverse.h — wave64 two-level BVH traversal (closest-hit and any-hit).
//
// Restates, with the reference's exact visiting order and culling rules,
//   IntersectTopBVH   raytracer.cpp:527-756     (closest hit, ANYHIT = false)
//   InShadow's loop   raytracer.cpp:1030-1243   (first accepted hit, ANYHIT = true)
//   IntersectMeshBVH  raytracer.cpp:758-811
//   IntersectAABB     raytracer.cpp:815-894
//   IntersectsTriangle_Bary raytracer.cpp:896-947, IntersectSphere :949-980
//
// Differences in FORM, none in result:
//   * an interior node's record holds both children's boxes, so both are tested
//     when the parent is visited.  The left child is what the reference pops
//     next, with the same minT, so its test is the reference's test.  The right
//     child is popped by the reference only after the whole left subtree; its
//     box entry distance tEnter is kept on the stack and re-checked against the
//     then-current minT at pop time.  Since IntersectAABB's verdict is
//     "min(minT, exit) < enter  or  (enter == 0 and min(minT, exit) <= 0)" and
//     the box was a hit at push time, the pop-time verdict is exactly
//     "minT < tEnter or (tEnter == 0 and minT <= 0)".
//   * both BVH levels run in ONE loop (a lane switches its active ray to object
//     space and back), so lanes at different levels share the node-fetch /
//     box-test instructions instead of serialising nested loops.
//   * 1/dir is computed once per active ray (the reference recomputes the same
//     quotient at every box, raytracer.cpp:822,842,862).
#pragma once
#include "rt_device.h"

#define RT_STACK_LDS 16          // entries per lane kept in LDS
#define RT_STACK_MAX 64          // the reference's stack[64] (raytracer.cpp:533,766)
#define RT_TRACE_BLOCK 256       // 4 waves per workgroup

struct TraceStack {
    uint32_t* lds_link;   // [RT_STACK_LDS][64] for this wave
    float*    lds_t;      // [RT_STACK_LDS][64]
    uint2*    spill;      // [(RT_STACK_MAX-RT_STACK_LDS)][64] for this wave (global)
    uint32_t  lane;
    uint32_t  sp;
    uint32_t  overflow;

    __device__ __forceinline__ void push(uint32_t link, float t) {
        if (sp < RT_STACK_LDS) {
            lds_link[sp * 64 + lane] = link;
            lds_t[sp * 64 + lane] = t;
        } else if (sp < RT_STACK_MAX) {
            spill[(sp - RT_STACK_LDS) * 64 + lane] = make_uint2(link, __float_as_uint(t));
        } else {
            overflow = 1;   // the reference's fixed array would overflow here as well
            return;
        }
        sp++;
    }
    // Pop the next entry the reference would not cull (see file header).
    __device__ __forceinline__ uint32_t pop(float tmax) {
        while (sp > 0) {
            --sp;
            uint32_t link;
            float t;
            if (sp < RT_STACK_LDS) {
                link = lds_link[sp * 64 + lane];
                t = lds_t[sp * 64 + lane];
            } else {
                uint2 e = spill[(sp - RT_STACK_LDS) * 64 + lane];
                link = e.x;
                t = __uint_as_float(e.y);
            }
            if (!((tmax < t) || (t == 0.0f && tmax <= 0.0f))) return link;
        }
        return RT_LINK_DONE;
    }
};

// The ray a lane is currently traversing with (world or object space).
struct ActiveRay {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;     // 1.0f / d (IEEE), unused where d == 0
    uint32_t zmask;       // bit a set: d[a] == 0 (raytracer.cpp:821,841,861 else-branches)
    __device__ __forceinline__ void set(V3 o, V3 d) {
        ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
        ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z;
        zmask = (d.x == 0.0f ? 1u : 0u) | (d.y == 0.0f ? 2u : 0u) | (d.z == 0.0f ? 4u : 0u);
    }
};

// One axis of IntersectAABB (raytracer.cpp:821-838).  std::max(a,b) = a<b?b:a
// and std::min(a,b) = b<a?b:a keep the first argument when the second is NaN.
__device__ __forceinline__ void slab(float bmin, float bmax, float o, float inv, bool dzero, float& tMin, float& tMax) {
    float lo, hi;
    if (!dzero) {
        float t0 = (bmin - o) * inv;
        float t1 = (bmax - o) * inv;
        bool sw = t1 < t0;
        lo = sw ? t1 : t0;
        hi = sw ? t0 : t1;
    } else {
        bool outside = (o < bmin) || (o > bmax);
        lo = outside ? RT_INF : -RT_INF;
        hi = outside ? -RT_INF : RT_INF;
    }
    tMin = (tMin < lo) ? lo : tMin;
    tMax = (hi < tMax) ? hi : tMax;
}

// IntersectAABB verdict + entry distance.  The per-axis early returns of the
// reference are equivalent to one final test because tMin only grows and tMax
// only shrinks; the trailing `tHit > minT` test (raytracer.cpp:889) can never
// fire once tMax <= minT and tMin <= tMax.
__device__ __forceinline__ bool box_test(const ActiveRay& r, float mnx, float mny, float mnz, float mxx, float mxy,
                                         float mxz, float tmax, float& tEnter) {
    float tMin = 0.0f, tMax = tmax;
    if (r.zmask == 0) {
        slab(mnx, mxx, r.ox, r.ix, false, tMin, tMax);
        slab(mny, mxy, r.oy, r.iy, false, tMin, tMax);
        slab(mnz, mxz, r.oz, r.iz, false, tMin, tMax);
    } else {
        slab(mnx, mxx, r.ox, r.ix, (r.zmask & 1u) != 0, tMin, tMax);
        slab(mny, mxy, r.oy, r.iy, (r.zmask & 2u) != 0, tMin, tMax);
        slab(mnz, mxz, r.oz, r.iz, (r.zmask & 4u) != 0, tMin, tMax);
    }
    tEnter = tMin;
    bool miss = (tMax < tMin) || (!(tMin > 0.0f) && tMax <= 0.0f);
    return !miss;
}

// IntersectsTriangle_Bary, raytracer.cpp:896-947, with col_A0/col_A1 preloaded.
__device__ __forceinline__ bool tri_test(const ActiveRay& r, float ax, float ay, float az, float e1x, float e1y,
                                         float e1z, float e2x, float e2y, float e2z, float minT, float& t_out,
                                         float& beta_out, float& gamma_out) {
    float bx = ax - r.ox, by = ay - r.oy, bz = az - r.oz;
    float detA = det3x3(e1x, e2x, r.dx, e1y, e2y, r.dy, e1z, e2z, r.dz);
    if (fabsf(detA) < RT_EPS_PARALLEL) return false;
    float beta = det3x3(bx, e2x, r.dx, by, e2y, r.dy, bz, e2z, r.dz) / detA;
    if (beta < -RT_EPS_PARALLEL) return false;
    float gamma = det3x3(e1x, bx, r.dx, e1y, by, r.dy, e1z, bz, r.dz) / detA;
    if ((gamma < -RT_EPS_PARALLEL) || ((beta + gamma) > (1.0f + RT_EPS_PARALLEL))) return false;
    float t = det3x3(e1x, e2x, bx, e1y, e2y, by, e1z, e2z, bz) / detA;
    if (t < -RT_EPS_PARALLEL || t >= minT) return false;
    t_out = t;
    beta_out = beta;
    gamma_out = gamma;
    return true;
}

// IntersectSphere, raytracer.cpp:949-980 (sqrt(double(delta)) narrowed to float
// is the correctly rounded sqrtf: 53 >= 2*24+2).
__device__ __forceinline__ bool sphere_test(const ActiveRay& r, float cx, float cy, float cz, float radius, float minT,
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
__device__ __forceinline__ bool plane_test(V3 o, V3 d, const DPlane& p, float minT, float& t_out) {
    V3 n = mk(p.nx, p.ny, p.nz);
    float den = dot(d, n);
    if (fabsf(den) < RT_EPS_PARALLEL) return false;
    float t = -(dot(o, n) + p.d) / den;
    if (t < 0.0f || t >= minT) return false;
    t_out = t;
    return true;
}

struct HitOut {
    float    t;          // minT (world; see raytracer.cpp:605-609 for what "world t" means)
    uint32_t kind_index; // kind<<28 | object index, 0xFFFFFFFF = miss
    uint32_t tri;        // mesh: global DTri slot of the hit triangle
    float    beta, gamma;
};

// Per-object ray set-up, raytracer.cpp:559-587 (and :1070-1095 for shadows).
__device__ __forceinline__ void object_space_ray(const DObject& ob, V3 ro, V3 rd, float time, float minT, V3& to,
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
__device__ __forceinline__ float world_t(const DObject& ob, V3 ro, float time, V3 to, V3 td, float t) {
    if (!(ob.flags & RT_OBJ_HAS_TRANSFORM)) return t;
    V3 objHit = to + td * t;
    V3 worldHit = xf_point(ob.M, objHit);
    if (ob.flags & RT_OBJ_HAS_MOTION_BLUR) worldHit = worldHit + mk(ob.mbx, ob.mby, ob.mbz) * time;
    return vlen(worldHit - ro);
}

// Traverse the scene's BVHs.  On entry `minT`/`has`/`out` carry the result of
// the plane loop (FindClosestHit raytracer.cpp:423-435 / InShadow :1022-1028).
// ANYHIT: returns true at the first accepted hit (InShadow).  Otherwise updates
// out/minT and returns `has`.
template <bool ANYHIT>
__device__ bool traverse(const DScene& S, V3 ro, V3 rd, float time, float minT, bool has, HitOut& out,
                         TraceStack& st) {
    st.sp = 0;
    ActiveRay ar;
    ar.set(ro, rd);
    float tmax = minT;           // culling bound of the active level
    bool in_mesh = false;
    uint32_t top_slot = 0, top_remaining = 0;
    // state of the mesh traversal in flight
    uint32_t cur_obj = 0;
    V3 obj_o = ro, obj_d = rd;
    bool mesh_found = false;
    uint32_t mesh_tri = 0;
    float mesh_beta = 0.f, mesh_gamma = 0.f;
    uint32_t cur_kind_index = 0;

    uint32_t cur = RT_LINK_DONE;
    if (S.top_root_link != RT_LINK_DONE) {
        float te;
        if (box_test(ar, S.top_rmin[0], S.top_rmin[1], S.top_rmin[2], S.top_rmax[0], S.top_rmax[1], S.top_rmax[2], tmax, te))
            cur = S.top_root_link;
    }

    for (;;) {
        // ---- interior nodes of either level
        while (!(cur & RT_LINK_LEAF)) {
            const DNode* n = &S.nodes[cur];
            const float4 a = n->a, b = n->b, c = n->c;
            const uint32_t ll = n->llink, rl = n->rlink;
            float tl, tr;
            const bool hl = box_test(ar, a.x, a.y, a.z, a.w, b.x, b.y, tmax, tl);
            const bool hr = box_test(ar, b.z, b.w, c.x, c.y, c.z, c.w, tmax, tr);
            if (hl) {
                if (hr) st.push(rl, tr);
                cur = ll;
            } else if (hr) {
                cur = rl;
            } else {
                cur = st.pop(tmax);
            }
        }

        if (in_mesh) {
            if (cur != RT_LINK_SENTINEL) {
                // ---- mesh leaf: triangles in primIdx order (raytracer.cpp:781-798)
                uint32_t ti = RT_LINK_INDEX(cur);
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(&S.tris[ti]);
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    float t, bb, gg;
                    if (tri_test(ar, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, tmax, t, bb, gg)) {
                        tmax = t;
                        mesh_found = true;
                        mesh_tri = ti;
                        mesh_beta = bb;
                        mesh_gamma = gg;
                        if (ANYHIT) {
                            // Untransformed mesh: worldT == t < testMinT == minT, so the
                            // reference's verdict is already known.
                            if (!(S.objects[cur_obj].flags & RT_OBJ_HAS_TRANSFORM)) return true;
                        }
                    }
                    if (__float_as_uint(q2.z) <= 1u) break;   // remaining
                    ti++;
                }
                cur = st.pop(tmax);
                continue;
            }
            // ---- mesh traversal finished (IntersectMeshBVH returned), raytracer.cpp:596-620
            in_mesh = false;
            if (mesh_found) {
                const DObject& ob = S.objects[cur_obj];
                float wt = world_t(ob, ro, time, obj_o, obj_d, tmax);
                if (wt < minT) {
                    if (ANYHIT) return true;
                    minT = wt;
                    has = true;
                    out.kind_index = cur_kind_index;
                    out.tri = mesh_tri;
                    out.beta = mesh_beta;
                    out.gamma = mesh_gamma;
                }
            }
            ar.set(ro, rd);
            tmax = minT;
        } else {
            if (cur == RT_LINK_DONE) break;
            top_slot = RT_LINK_INDEX(cur);
            top_remaining = 1;   // refined from the first record below
        }

        // ---- prims of the current top-level leaf, in order (raytracer.cpp:548-745)
        bool entered = false;
        while (top_remaining != 0) {
            const DTopPrim tp = S.top_prims[top_slot];
            top_slot++;
            top_remaining = tp.remaining - 1;
            const DObject& ob = S.objects[tp.obj];
            V3 to, td;
            float testMinT;
            object_space_ray(ob, ro, rd, time, minT, to, td, testMinT);
            if (tp.kind == RT_KIND_MESH) {
                if (ob.root_link == RT_LINK_DONE) continue;
                ActiveRay mr;
                mr.set(to, td);
                float te;
                if (!box_test(mr, ob.rminx, ob.rminy, ob.rminz, ob.rmaxx, ob.rmaxy, ob.rmaxz, testMinT, te)) continue;
                ar = mr;
                tmax = testMinT;
                obj_o = to;
                obj_d = td;
                cur_obj = tp.obj;
                cur_kind_index = (RT_KIND_MESH << 28) | tp.index;
                mesh_found = false;
                in_mesh = true;
                st.push(RT_LINK_SENTINEL, -1.0f);
                cur = ob.root_link;
                entered = true;
                break;
            } else if (tp.kind == RT_KIND_SPHERE) {
                ActiveRay sr;
                sr.set(to, td);
                float t;
                if (sphere_test(sr, ob.p[0], ob.p[1], ob.p[2], ob.p[3], testMinT, t)) {
                    float wt = world_t(ob, ro, time, to, td, t);
                    if (wt < minT) {
                        if (ANYHIT) return true;
                        minT = wt;
                        has = true;
                        out.kind_index = (RT_KIND_SPHERE << 28) | tp.index;
                        out.tri = 0;
                        tmax = minT;
                    }
                }
            } else {   // RT_KIND_TRIANGLE
                ActiveRay tr;
                tr.set(to, td);
                float t, bb, gg;
                if (tri_test(tr, ob.p[0], ob.p[1], ob.p[2], ob.p[3], ob.p[4], ob.p[5], ob.p[6], ob.p[7], ob.p[8],
                             testMinT, t, bb, gg)) {
                    float wt = world_t(ob, ro, time, to, td, t);
                    if (wt < minT) {
                        if (ANYHIT) return true;
                        minT = wt;
                        has = true;
                        out.kind_index = (RT_KIND_TRIANGLE << 28) | tp.index;
                        out.tri = 0;
                        tmax = minT;
                    }
                }
            }
        }
        if (!entered) cur = st.pop(tmax);
    }
    out.t = minT;
    return ANYHIT ? false : has;
}
