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
//   * one node record holds the boxes of the four GRANDCHILDREN of a reference
//     interior node (two levels collapsed), in the reference's depth-first
//     order, and all four are tested when the record is visited.  The first hit
//     slot is what the reference reaches next with the same minT.  Every other
//     hit slot is reached by the reference only after the subtrees before it;
//     its box entry distance tEnter is kept on the stack and re-checked against
//     the then-current minT at pop time.  Since IntersectAABB's verdict is
//     "min(minT, exit) < enter  or  (enter == 0 and min(minT, exit) <= 0)" and
//     the box was a hit at push time, the pop-time verdict is exactly
//     "minT < tEnter or (tEnter == 0 and minT <= 0)".  The intermediate node
//     between a record and its slots needs no test of its own: its box contains
//     the slot's box, every slab operation is monotone under rounding, and the
//     reference tests parent and child with the same minT (no primitive test in
//     between), so "child box hit" implies "parent box hit" bit for bit; and a
//     parent whose children all miss has no side effect in the reference.
//   * mesh leaves are POSTPONED: a lane that reaches a leaf records (leaf, tEnter)
//     in a small per-lane queue and keeps traversing; when its queue is full or
//     its traversal is exhausted it waits, and once every lane of the wave waits
//     the queued leaves are tested, in order, each first re-checked against the
//     current minT with the pop-time rule above.  This is exact, not approximate:
//     a leaf L is tested by the reference iff L's own box passes IntersectAABB
//     against the minT left by the leaves before it (every ancestor's box
//     contains L's box and is tested no later with a minT no smaller, so by
//     monotonicity it passes whenever L's does), and traversing with a stale,
//     larger minT can only add candidates, which that re-check removes again.
//     It lifts SIMD utilisation of the node loop from 21 % to ~55 %.
//   * a direction component that is exactly 0 takes 1/d = +inf instead of the
//     reference's separate branch (raytracer.cpp:834-838): (b-o)*inf is -inf,
//     NaN or +inf for o inside, on, or outside a face, and NaN operands are
//     ignored by the reference's own std::max/std::min argument order, which
//     reproduces "miss iff o < min or o > max" exactly, branch-free.
//   * both BVH levels run in ONE loop (a lane switches its active ray to object
//     space and back), so lanes at different levels share the node-fetch /
//     box-test instructions instead of serialising nested loops.
//   * 1/dir is computed once per active ray (the reference recomputes the same
//     quotient at every box, raytracer.cpp:822,842,862).
#pragma once
#include "rt_device.h"

#ifndef RT_STACK_LDS
#define RT_STACK_LDS 12          // entries per lane kept in LDS
#endif
#ifndef RT_LEAF_QUEUE
#define RT_LEAF_QUEUE 8          // mesh leaves a lane may postpone before the wave switches to triangle tests
#endif
#ifndef RT_LEAF_QUEUE_ANY
#define RT_LEAF_QUEUE_ANY 4      // same for any-hit (shadow) rays: the first accepted hit ends the ray, so postpone less
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4           // __launch_bounds__ waves per SIMD for the trace kernels
#endif
#define RT_STACK_MAX 64          // the reference's stack[64] (raytracer.cpp:533,766)
#define RT_TRACE_BLOCK 256       // 4 waves per workgroup

// Diagnostic build only (-DRT_TRACE_STATS): per-wave step counts to measure SIMD
// utilisation of the traversal loop.  Never compiled into the shipped library.
#ifdef RT_TRACE_STATS
struct TraceStatsAcc {
    unsigned long long wave_node_steps, lane_node_steps, wave_tri_steps, lane_tri_steps, wave_outer, lane_rays, wave_batches, wave_pops, lane_pops;
};
#define RT_STAT(x) x
#else
#define RT_STAT(x)
#endif

struct TraceStack {
    uint32_t* lds_link;   // [RT_STACK_LDS][64] for this wave
    float*    lds_t;      // [RT_STACK_LDS][64]
    uint2*    spill;      // [(RT_STACK_MAX-RT_STACK_LDS)][64] for this wave (global)
    uint32_t* q_link;     // [RT_LEAF_QUEUE][64] postponed mesh leaves, in visiting order
    float*    q_t;        // [RT_LEAF_QUEUE][64] their box entry distances
    uint32_t  lane;
    uint32_t  sp;
    uint32_t  overflow;
#ifdef RT_TRACE_STATS
    TraceStatsAcc* acc;
#endif

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
    __device__ __forceinline__ uint32_t pop(float tmax, float& t_out) {
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
            if (!((tmax < t) || (t == 0.0f && tmax <= 0.0f))) { t_out = t; return link; }
        }
        return RT_LINK_DONE;
    }
};

// The ray a lane is currently traversing with (world or object space).
struct ActiveRay {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;     // 1.0f / d (IEEE); +inf where d == 0 (see file header)
    __device__ __forceinline__ void set(V3 o, V3 d) {
        ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
        ix = (d.x == 0.0f) ? RT_INF : 1.0f / d.x;
        iy = (d.y == 0.0f) ? RT_INF : 1.0f / d.y;
        iz = (d.z == 0.0f) ? RT_INF : 1.0f / d.z;
    }
};

// IntersectAABB (raytracer.cpp:815-894) verdict + entry distance, branch-free.
// Per axis: t0,t1, swapped iff t1 < t0 (as the reference's explicit swap: a NaN
// never swaps); tMin = std::max(tMin, lo), tMax = std::min(tMax, hi) keep their
// first argument when the second is NaN, which is what fmaxf/fminf (v_max_f32 /
// v_min_f32) do since tMin/tMax themselves are never NaN.  The per-axis early
// returns equal one final test because tMin only grows and tMax only shrinks;
// "hit" = !(tMax < tMin) && !(tMin == 0 && tMax <= 0) = (tMin <= tMax) && (tMax > 0);
// the trailing `tHit > minT` test (raytracer.cpp:889) cannot fire once tMax <= minT.
__device__ __forceinline__ bool box_test(const ActiveRay& r, float mnx, float mny, float mnz, float mxx, float mxy,
                                         float mxz, float tmax, float& tEnter) {
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

__device__ __forceinline__ uint32_t leaf_count(const DScene& S, uint32_t link, bool top) {
    uint32_t c = RT_LEAF_COUNT(link);
    if (c != RT_LEAF_BIG) return c;
    // unsplittable node with more than 30 prims (raytracer.cpp:183,317): binary search the side table
    const uint32_t key = RT_LEAF_FIRST(link) | (top ? 0x80000000u : 0u);
    uint32_t lo = 0, hi = S.n_big_leaves;
    while (lo + 1 < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (S.big_leaves[mid].first <= key) lo = mid; else hi = mid;
    }
    return S.big_leaves[lo].count;
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
    bool mesh_found = false;
    uint32_t mesh_tri = 0;
    float mesh_beta = 0.f, mesh_gamma = 0.f;
    uint32_t cur_kind_index = 0;
    uint32_t qn = 0;             // postponed leaves in the queue
    constexpr uint32_t QCAP = ANYHIT ? RT_LEAF_QUEUE_ANY : RT_LEAF_QUEUE;
    static_assert(RT_LEAF_QUEUE_ANY <= RT_LEAF_QUEUE, "the LDS queue is sized by RT_LEAF_QUEUE");

    uint32_t cur = RT_LINK_DONE;
    float cur_t = 0.f;           // box entry distance of `cur`
    if (S.top_root_link != RT_LINK_DONE) {
        float te;
        if (box_test(ar, S.top_rmin[0], S.top_rmin[1], S.top_rmin[2], S.top_rmax[0], S.top_rmax[1], S.top_rmax[2], tmax, te)) {
            cur = S.top_root_link;
            cur_t = te;
        }
    }

    for (;;) {
        // ---- node phase: collapsed interior records of either level; mesh leaves are queued
        RT_STAT(if (st.lane == __ffsll((long long)__ballot(1)) - 1) st.acc->wave_outer++;)
        for (;;) {
            // A lane can make progress here while it sits on an interior record, or on a mesh
            // leaf it still has queue room for.  The phase ends for the WAVE when no lane can
            // (wave-uniform exit: the two phases must not be interleaved by the compiler).
            const bool leaf_here = (cur & RT_LINK_LEAF) != 0;
            const bool want = !leaf_here || (in_mesh && cur != RT_LINK_SENTINEL && qn < QCAP);
            if (!__any(want)) break;
            if (want) {
                if (leaf_here) {
                    st.q_link[qn * 64 + st.lane] = cur;
                    st.q_t[qn * 64 + st.lane] = cur_t;
                    qn++;
                    cur = st.pop(tmax, cur_t);
                }
                if (!(cur & RT_LINK_LEAF)) {
                    RT_STAT({ unsigned long long m = __ballot(1); if (st.lane == __ffsll((long long)m) - 1) { st.acc->wave_node_steps++; st.acc->lane_node_steps += __popcll(m); } })
                    const float4* p = reinterpret_cast<const float4*>(&S.nodes[cur]);
                    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5];
                    const uint4 lk = reinterpret_cast<const uint4*>(p)[6];
                    const uint32_t mask = reinterpret_cast<const uint32_t*>(p)[28];
                    float t0, t1, t2, t3;
                    const bool h0 = box_test(ar, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tmax, t0);                 // slot 0 is always valid
                    const bool h1 = box_test(ar, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tmax, t1) && (mask & 2u);
                    const bool h2 = box_test(ar, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, tmax, t2) && (mask & 4u);
                    const bool h3 = box_test(ar, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, tmax, t3) && (mask & 8u);
                    // visit order = slot order; later hits are pushed last-first so they pop in order
                    uint32_t nx = RT_LINK_NONE;
                    float tn = 0.f;
                    if (h3) { nx = lk.w; tn = t3; }
                    if (h2) { if (nx != RT_LINK_NONE) st.push(nx, tn); nx = lk.z; tn = t2; }
                    if (h1) { if (nx != RT_LINK_NONE) st.push(nx, tn); nx = lk.y; tn = t1; }
                    if (h0) { if (nx != RT_LINK_NONE) st.push(nx, tn); nx = lk.x; tn = t0; }
                    if (nx != RT_LINK_NONE) { cur = nx; cur_t = tn; }
                    else cur = st.pop(tmax, cur_t);
                }
            }
        }

        // ---- triangle phase: the postponed leaves, in visiting order (raytracer.cpp:781-798)
        for (uint32_t qi = 0; qi < qn; qi++) {
            const uint32_t link = st.q_link[qi * 64 + st.lane];
            const float te = st.q_t[qi * 64 + st.lane];
            if ((tmax < te) || (te == 0.0f && tmax <= 0.0f)) continue;   // the reference would cull this leaf's box now
            uint32_t ti = RT_LEAF_FIRST(link);
            uint32_t n = leaf_count(S, link, false);
            for (; n != 0; n--, ti++) {
                RT_STAT({ unsigned long long m = __ballot(1); if (st.lane == __ffsll((long long)m) - 1) { st.acc->wave_tri_steps++; st.acc->lane_tri_steps += __popcll(m); } })
                const float* tp = reinterpret_cast<const float*>(&S.tris[ti]);
                float t, bb, gg;
                if (tri_test(ar, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6], tp[7], tp[8], tmax, t, bb, gg)) {
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
            }
        }
        qn = 0;

        if (in_mesh) {
            if (cur != RT_LINK_SENTINEL) {
                // a mesh leaf is waiting (queue was full): it was accepted with an older minT, re-check it
                if ((tmax < cur_t) || (cur_t == 0.0f && tmax <= 0.0f)) cur = st.pop(tmax, cur_t);
                continue;
            }
            // ---- mesh traversal finished (IntersectMeshBVH returned), raytracer.cpp:596-620
            in_mesh = false;
            if (mesh_found) {
                const DObject& ob = S.objects[cur_obj];
                float wt = world_t(ob, ro, time, mk(ar.ox, ar.oy, ar.oz), mk(ar.dx, ar.dy, ar.dz), tmax);
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
            top_slot = RT_LEAF_FIRST(cur);
            top_remaining = leaf_count(S, cur, true);
        }

        // ---- prims of the current top-level leaf, in order (raytracer.cpp:548-745)
        bool entered = false;
        while (top_remaining != 0) {
            const DTopPrim tp = S.top_prims[top_slot];
            top_slot++;
            top_remaining--;
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
                cur_obj = tp.obj;
                cur_kind_index = (RT_KIND_MESH << 28) | tp.index;
                mesh_found = false;
                in_mesh = true;
                st.push(RT_LINK_SENTINEL, -1.0f);
                cur = ob.root_link;
                cur_t = te;
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
        if (!entered) cur = st.pop(tmax, cur_t);
    }
    out.t = minT;
    return ANYHIT ? false : has;
}
