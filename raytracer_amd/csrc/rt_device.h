// rt_device.h — device data layout and FP32 math for the gfx950 ray path.
//
// Arithmetic contract: IEEE binary32, compiled with
// -ffp-contract=off, correctly-rounded divide and sqrt (hipcc default), operation
// order exactly as in the reference lines cited at each function, so that every
// intermediate equals the x86-64 (no-FMA) reference value bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt.h"

#define RT_INF __builtin_huge_valf()
#define RT_FLT_MAX 3.402823466e+38F   // MathF.h:6
#define RT_EPS_PARALLEL 1e-8f         // raytracer.h:12

// ------------------------------------------------------------ link encoding
// A link names the next thing to visit.
//   bit31 clear : index of a DNode4.
//   bit31 set   : a leaf.  bits 30..26 = prim count (1..30; 31 = "big leaf", the
//                 count is in DScene::big_leaves), bits 25..0 = index of the first
//                 DTri (mesh level) or DTopPrim slot (top level).
//   two reserved values mark the bottom of the stack segments.
#define RT_LINK_LEAF      0x80000000u
#define RT_LINK_DONE      0xFFFFFFFFu   // empty stack at the top level / empty BVH
#define RT_LINK_SENTINEL  0xFFFFFFFEu   // bottom of a mesh traversal's stack segment
#define RT_LINK_NONE      0xFFFFFFFDu   // "no slot hit" marker inside the node step
#define RT_LEAF_FIRST(l)  ((l) & 0x03FFFFFFu)
#define RT_LEAF_COUNT(l)  (((l) >> 26) & 31u)
#define RT_LEAF_MAX_FIRST 0x03FFFFF0u
#define RT_LEAF_BIG       31u
#define RT_NODE_RECORDS_MAX 0x01FFFFF0u   // node records a scene may have: their byte offsets (x 128) fit 32 bits (rt_traverse.h)

// Two levels of a reference BVH (Bvh.h:39-49) collapsed into one 128-byte
// record: for a reference interior node n with children L, R the four slots are
// L.left, L.right, R.left, R.right in that (= the reference's depth-first)
// order; a child that is itself a leaf occupies its first slot and leaves the
// second one empty (mask bit clear).  Planes by axis: f[0..3] = min.x of the four slots, f[4..7] = max.x,
// f[8..11] = min.y, f[12..15] = max.y, f[16..19] = min.z, f[20..23] = max.z, so that a lane fetches the four
// planes its ray meets first on an axis with one 16-byte load picked by the sign of 1/d.
struct __attribute__((aligned(128))) DNode4 {
    float    f[24];
    uint32_t link[4];
    uint32_t mask;
    uint32_t pad[3];
};
static_assert(sizeof(DNode4) == 128, "DNode4 must be 128 bytes");

// Leaf-ordered triangle record: position p of a mesh BVH's primIdx array
// (raytracer.cpp:783-789 resolved at upload: primIdx -> prims[].faceIdx ->
// faces[] -> vertex_data[]).  e1 = a-b, e2 = a-c are the first two columns of
// the Cramer system (raytracer.cpp:902-903), computed once on the host with the
// same FP32 subtraction.  36 bytes.
struct DTri {
    float ax, ay, az, e1x, e1y, e1z, e2x, e2y, e2z;
};
static_assert(sizeof(DTri) == 36, "DTri is 36 bytes");

struct DBigLeaf { uint32_t first, count; };   // leaves with more than 30 prims (unsplittable nodes)

// Shading-side data of the same triangle slot (only read for the closest hit).
struct __attribute__((aligned(16))) DTriShade {
    float nx, ny, nz;     // Face::n_unit
    int32_t i0, i1, i2;   // 1-based vertex ids (smooth normals)
    uint32_t face;        // faceIdx inside the mesh's face set (hit identity)
    uint32_t pad;
};

// Object record shared by mesh instances, spheres and triangles.
struct __attribute__((aligned(16))) DObject {
    float    M[16];       // transformation (row-major)
    float    Minv[16];    // invTransformation
    float    mbx, mby, mbz;
    uint32_t flags;       // RT_OBJ_* | DOBJ_*
    int32_t  material_id;
    uint32_t root_link;   // mesh: link of the BVH root (RT_LINK_DONE if empty)
    float    rminx, rminy, rminz, rmaxx, rmaxy, rmaxz;   // mesh: root box
    // sphere: p0..p3 = centre.xyz, radius.  triangle: a, e1, e2, n_unit in p0..p11
    float    p[12];
};
#define DOBJ_NEG_DET (1u << 8)   // transformation.determinant3x3() < 0 (raytracer.cpp:468,489,505)

// One slot of the top-level leaf order (topPrimIdx, raytracer.cpp:550-551).
struct DTopPrim {
    uint32_t kind;        // RT_KIND_*
    uint32_t index;       // object index inside its kind
    uint32_t remaining;   // prims left in this leaf, this one included
    uint32_t obj;         // index into DScene::objects
};

// The top level as the reference walks it (IntersectTopBVH, raytracer.cpp:526-745: a stack, left child before right, each
// node's box tested when it is popped): the nodes of the top BVH in pre-order, every leaf followed by its prims.  An item
// is 32 bytes (two uint4).  Node: words 0-5 bmin.xyz, bmax.xyz, word 6 the index of the item after the node's subtree
// (where the walk goes on when the box is missed), word 7 RT_TOPITEM_NODE.  Prim: words 0-2 kind, index inside its kind,
// index into DScene::objects, word 7 RT_TOPITEM_PRIM.  A lane's whole top-level state is one item index.
#define RT_TOPITEM_NODE 0u
#define RT_TOPITEM_PRIM 1u
// RT_INLINE_LEAF_MESH: an instance of a mesh whose BVH is one leaf of at most RT_INLINE_LEAF_MAX triangles, without
// transformation or motion blur (a ground quad, a wall), is written as a node item with the mesh's root box followed by one
// RT_KIND_MESHTRI item per triangle (word 3: the triangle's slot in DScene::tris).  For such an instance the reference's
// IntersectMeshBVH (raytracer.cpp:758-811) is: the root box against minT, then each triangle against the running bound, then
// `t < minT` — which is what the walk does with these items, on the same operands (testRay == ray, testMinT == minT, worldT == t).
#ifndef RT_INLINE_LEAF_MESH
#define RT_INLINE_LEAF_MESH 1
#endif
#define RT_INLINE_LEAF_MAX 4
#define RT_KIND_MESHTRI 4u
#define RT_LINK_TOPSCAN 0xFFFFFFFCu   // `cur` of a ray that is between objects: its place is the item index, not a link

struct DMaterial {
    uint32_t type;
    float ka[3], kd[3], ks[3], km[3];
    float phong, ior, absorb_index;
    float absorb_coef[3];
    float roughness;
};

struct DPlane { float nx, ny, nz, d; int32_t material_id; };
struct DPointLight { float px, py, pz, ix, iy, iz; };

struct DScene {
    const DNode4*     nodes;
    const DTri*       tris;
    const DTriShade*  tri_shade;
    const float4*     vnormals;      // vertex normal per vertex_data entry (w unused)
    const uint32_t*   tri_rank;      // NULL, or per triangle record its position in the reference's leaf order (rebuilt BVHs, rt_rebuild.h)
    const DObject*    objects;
    const DTopPrim*   top_prims;
    const uint4*      top_items;     // 2 per item (see RT_TOPITEM_*)
    uint32_t          n_top_items;
    const DPlane*     planes;
    const DMaterial*  materials;
    const DPointLight* point_lights;
    const DBigLeaf*   big_leaves;    // sorted by first; mesh and top-level leaves share the table (top entries have bit31 of first set)
    uint32_t n_big_leaves;
    uint32_t stack_cap;   // traversal stack entries per ray (>= RT_STACK_MIN_CAP; covers the BVH depths of this scene)
    uint32_t donate_ok;   // every mesh BVH visits its leaves in ascending triangle-slot order (needed by subtree donation)
    uint32_t n_planes, n_point_lights, n_materials;
    uint32_t top_root_link;
    float    top_rmin[3], top_rmax[3];
    float    background[3], ambient[3];
    float    shadow_eps;
    int32_t  max_depth;
    uint32_t mesh_obj_base, sphere_obj_base, tri_obj_base;   // objects[] = meshes | spheres | triangles
};

struct DCamera {
    float pos[3], u[3], v[3], w[3], q[3];
    float pixel_w, pixel_h;
    int32_t image_w, image_h, num_samples, sps;
    float aperture, focus;
    int32_t dof;
};

// ------------------------------------------------------------------- Vec3f.h
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ldv(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // Vec3f.h:16-18
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                   // Vec3f.h:20-24
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vlen(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }   // Vec3f.h:26-28
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 emul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 vdiv(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }

// MathF.h:18-28
__device__ __forceinline__ float q_rsqrt(float x) {
    if (x <= 0.0f) return 0.0f;
    uint32_t i = __float_as_uint(x);
    i = 0x5f3759dfu - (i >> 1);
    float y = __uint_as_float(i);
    y = y * (1.5f - 0.5f * x * y * y);
    return y;
}
// Vec3f.h:30-39
__device__ __forceinline__ V3 normalize(V3 a) {
    float inv = q_rsqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (inv == 0.0f) return mk(0.f, 0.f, 0.f);
    return a * inv;
}
// MathF.h:30-38
__device__ __forceinline__ float det3x3(float a11, float a12, float a13, float a21, float a22, float a23, float a31,
                                        float a32, float a33) {
    return a11 * (a22 * a33 - a23 * a32) - a12 * (a21 * a33 - a23 * a31) + a13 * (a21 * a32 - a22 * a31);
}
// Mat4f.h:21-27
__device__ __forceinline__ V3 xf_point(const float* m, V3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1;
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1;
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1;
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * 1;
    // (x / 1.0f is x, bit for bit, for every x: with the affine matrices the parser composes w is exactly 1 and the three
    // IEEE divisions — the reference makes them, Mat4f.h — are skipped by a branch no lane takes)
    if (w != 1.0f) { x = x / w; y = y / w; z = z / w; }
    return mk(x, y, z);
}
// Mat4f.h:30-36
__device__ __forceinline__ V3 xf_vector(const float* m, V3 v) {
    float x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 0;
    float y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 0;
    float z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 0;
    return mk(x, y, z);
}
// transpose().transformVector(v): Mat4f.h:51-59 then 30-36
__device__ __forceinline__ V3 xf_vector_transposed(const float* m, V3 v) {
    float x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * 0;
    float y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * 0;
    float z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * 0;
    return mk(x, y, z);
}

// std::pow(float,float) / std::exp(float) of the reference (raytracer.cpp:1517, 1428, 1356-1358): glibc 2.35's
// powf / expf restated bit for bit (rt_libm.h; checked against the C library on the host by tests/libm_logic.cpp).
// OCML's powf / expf differ from glibc's by an ulp now and then, which the truncating byte conversion
// (raytracer.cpp:83-85) turns into a different byte about once per 10^5 shaded pixels.
#include "rt_libm.h"
__device__ __forceinline__ float rt_powf(float a, float b) { return rt_glibc_powf(a, b); }
__device__ __forceinline__ float rt_expf(float a) { return rt_glibc_expf(a); }
