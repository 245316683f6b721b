/*
 * rt.h — C ABI of the MI355X-native per-pixel ray/scene intersection path.
 *
 * This is the drop-in boundary for fatih-ozdal/Raytracer's pixel loop
 * (reference raytracer.cpp:40-87).  The reference has no FFI of its own: the
 * hot path is called from exactly one place, the OpenMP loop body in main().
 * A maintainer keeps parser::loadFromJson / BuildAllMeshBVHs / BuildTopLevelBVH
 * (raytracer.cpp:22-26) and stbi_write_png (raytracer.cpp:89) as they are,
 * flattens `Scene` + the BVH globals (raytracer.cpp:8-11, Bvh.h:76) into an
 * rt_scene_desc, and replaces the loop with rt_upload_scene + rt_render_tile.
 * The flatten step is raytracer_amd/host/rt_flatten.hpp; INTEGRATION.md shows
 * the replacement main().
 *
 * Conventions
 *   - plain C, POD structs, pointers + counts; all floats are IEEE binary32,
 *     all ids in the scene are 1-based exactly as in the reference
 *     (vertex_data[i-1], materials[id-1]).
 *   - every function returns 0 (RT_OK) or a negative rt_status; nothing throws
 *     or aborts across the boundary; rt_last_error() gives a thread-local
 *     message (the reference's only status convention is main() returning
 *     -1/0, raytracer.cpp:15-18,93).
 *   - caller owns all input arrays; rt_upload_scene copies and retains nothing.
 *   - there is NO CPU fallback in librt_hip.so: without a HIP device every
 *     entry point fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_H
#define RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum rt_status {
    RT_OK               =  0,
    RT_ERR_INVALID      = -1,  /* bad argument / inconsistent scene arrays       */
    RT_ERR_NO_DEVICE    = -2,  /* no HIP device or device index out of range     */
    RT_ERR_HIP          = -3,  /* a HIP runtime call failed (message has detail) */
    RT_ERR_OOM          = -4,  /* device allocation failed                       */
    RT_ERR_UNSUPPORTED  = -5,  /* scene feature outside the built scope          */
    RT_ERR_OVERFLOW     = -6,  /* ray pool / traversal stack limit exceeded      */
    RT_ERR_IO           = -7
} rt_status;

/* PrimKind, HitRecord.h:7-13 */
enum { RT_KIND_MESH = 0, RT_KIND_TRIANGLE = 1, RT_KIND_SPHERE = 2, RT_KIND_PLANE = 3 };
/* MaterialType, parser.h:57-63 */
enum { RT_MAT_NONE = 0, RT_MAT_MIRROR = 1, RT_MAT_CONDUCTOR = 2, RT_MAT_DIELECTRIC = 3 };

/* object flags (Mesh/Triangle/Sphere: parser.h:92-168) */
enum {
    RT_OBJ_HAS_TRANSFORM   = 1u << 0,   /* hasTransform                    */
    RT_OBJ_HAS_MOTION_BLUR = 1u << 1,   /* has_motion_blur                 */
    RT_OBJ_SMOOTH          = 1u << 2    /* Mesh::is_smooth                 */
};

/* ------------------------------------------------------ flattened scene POD */

/* BVHNode, Bvh.h:39-49 (32 B).  prim_count > 0 : leaf, left_or_first = first
 * index into the prim-index array; prim_count == 0 : interior, children are
 * left_or_first and left_or_first + 1 (raytracer.cpp:749-750). */
typedef struct rt_bvh_node {
    float   bmin[3];
    float   bmax[3];
    int32_t left_or_first;
    int32_t prim_count;
} rt_bvh_node;

/* TopPrim{kind,index}, Bvh.h:6-15 (bounds/centroid are build-only) */
typedef struct rt_top_prim {
    uint32_t kind;    /* RT_KIND_MESH / _TRIANGLE / _SPHERE */
    uint32_t index;   /* index into meshes / triangles / spheres */
} rt_top_prim;

/* Vertex, parser.h:79-83 */
typedef struct rt_vertex {
    float pos[3];
    float normal[3];
} rt_vertex;

/* Face, parser.h:85-90 */
typedef struct rt_face {
    int32_t i0, i1, i2;   /* 1-based into vertex_data */
    float   n_unit[3];
    float   plane_d;
} rt_face;

/* One MeshBVH (Bvh.h:62-73) together with the face array of the ORIGINAL mesh
 * it was built for (instances copy that vector, parser.cpp:725).  Offsets index
 * the concatenated arrays in rt_scene_desc. */
typedef struct rt_mesh_bvh {
    uint32_t node_offset;   /* into mesh_nodes; root is node_offset + 0          */
    uint32_t node_count;    /* nodesUsed                                         */
    uint32_t prim_offset;   /* into mesh_prim_face                               */
    uint32_t prim_count;    /* number of triangles                               */
    uint32_t face_offset;   /* into faces                                        */
    uint32_t face_count;
} rt_mesh_bvh;

/* Mesh, parser.h:92-121 — one per scene.meshes entry, instances included. */
typedef struct rt_mesh {
    int32_t  bvh_index;       /* Mesh::bvhIndex                                   */
    int32_t  material_id;     /* 1-based                                          */
    uint32_t flags;           /* RT_OBJ_*                                         */
    float    transform[16];   /* Mat4f row-major m[r][c]                          */
    float    inv_transform[16];
    float    motion_blur[3];
} rt_mesh;

/* Triangle, parser.h:123-144 */
typedef struct rt_triangle {
    int32_t  material_id;
    rt_face  face;
    uint32_t flags;
    float    transform[16];
    float    inv_transform[16];
    float    motion_blur[3];
} rt_triangle;

/* Sphere, parser.h:146-168 */
typedef struct rt_sphere {
    int32_t  material_id;
    int32_t  center_vertex_id;  /* 1-based */
    float    radius;
    uint32_t flags;
    float    transform[16];
    float    inv_transform[16];
    float    motion_blur[3];
} rt_sphere;

/* Plane, parser.h:170-190 (transform already baked in by parser.cpp:892-903) */
typedef struct rt_plane {
    int32_t material_id;
    float   n_unit[3];
    float   plane_d;
} rt_plane;

/* Material, parser.h:65-77 */
typedef struct rt_material {
    uint32_t type;              /* RT_MAT_* */
    float ambient_refl[3];
    float diffuse_refl[3];
    float specular_refl[3];
    float mirror_refl[3];
    float phong_exponent;
    float refraction_index;
    float absorption_index;
    float absorption_coef[3];
    float roughness;
} rt_material;

/* PointLight, parser.h:43-48 */
typedef struct rt_point_light {
    float position[3];
    float intensity[3];
} rt_point_light;

/* AreaLight, parser.h:50-55 */
typedef struct rt_area_light {
    float position[3];
    float normal[3];
    float size;
    float radiance[3];
} rt_area_light;

/* Scene (parser.h:192-207) + the BVH globals (raytracer.cpp:8-11, Bvh.h:76). */
typedef struct rt_scene_desc {
    /* scalars */
    float    background_color[3];
    float    ambient_light[3];
    float    shadow_ray_epsilon;
    int32_t  max_recursion_depth;

    /* top-level BVH: topBvhNodes[0..nodesUsed), topPrimIdx, topPrims */
    const rt_bvh_node* top_nodes;      uint32_t n_top_nodes;
    const uint32_t*    top_prim_idx;   uint32_t n_top_prims;
    const rt_top_prim* top_prims;      /* n_top_prims entries */

    /* mesh BVHs (concatenated) */
    const rt_mesh_bvh* mesh_bvhs;      uint32_t n_mesh_bvhs;
    const rt_bvh_node* mesh_nodes;     uint32_t n_mesh_nodes;
    /* for leaf slot p of a mesh BVH: faceIdx = prims[primIdx[p]].faceIdx
     * (raytracer.cpp:783-785), relative to that BVH's face_offset */
    const uint32_t*    mesh_prim_face; uint32_t n_mesh_prims;
    const rt_face*     faces;          uint32_t n_faces;

    const rt_vertex*   vertex_data;    uint32_t n_vertices;

    const rt_mesh*     meshes;         uint32_t n_meshes;
    const rt_triangle* triangles;      uint32_t n_triangles;
    const rt_sphere*   spheres;        uint32_t n_spheres;
    const rt_plane*    planes;         uint32_t n_planes;

    const rt_material*    materials;     uint32_t n_materials;
    const rt_point_light* point_lights;  uint32_t n_point_lights;
    const rt_area_light*  area_lights;   uint32_t n_area_lights;
} rt_scene_desc;

/* Camera fields the hot path reads (parser.h:21-41; raytracer.cpp:32-38,343-378).
 * image_width/height are the FULL image: the per-pixel RNG seed is
 * i*image_width + j with global i,j (raytracer.cpp:44). */
typedef struct rt_camera_desc {
    float   position[3];
    float   u[3], v[3], w[3];
    float   q[3];
    float   pixel_width, pixel_height;
    int32_t image_width, image_height;
    int32_t num_samples, samples_per_side;
    float   aperture_size, focus_distance;
    int32_t has_depth_of_field;
} rt_camera_desc;

/* Per-call counters and timings (all optional; zeroed by the callee). */
typedef struct rt_stats {
    uint64_t rays_primary;     /* FindClosestHit calls at depth 0               */
    uint64_t rays_secondary;   /* FindClosestHit calls at depth > 0             */
    uint64_t rays_shadow;      /* InShadow calls                                */
    uint64_t records_peak;     /* peak ray-record pool use                      */
    uint32_t chunks;           /* number of pixel chunks the tile was split in  */
    uint32_t launches_closest; /* launches of the closest-hit traversal kernel  */
    uint32_t launches_shadow;
    float    ms_total;         /* HIP-event time of the whole call on its stream */
    float    ms_raygen;
    float    ms_closest;       /* sum over launches of the closest-hit kernel    */
    float    ms_shade;
    float    ms_shadow;        /* sum over launches of the any-hit kernel        */
    float    ms_combine;       /* the bottom-up colour pass of the generations above 0 (0 on a depth-0 frame)  */
    float    ms_resolve;       /* generation 0's colours, the sum over a pixel's samples, clamp, RGB8          */
} rt_stats;

typedef struct rt_scene rt_scene;   /* opaque, library-owned, one per GPU */

/* render flags for rt_scene_set_option(RT_OPT_FLAGS) */
enum {
    RT_FLAG_TIMING = 1u << 0,  /* bracket kernels with HIP events -> rt_stats.ms_* */
    RT_FLAG_SEPARATE_LAUNCHES = 1u << 1,  /* trace the shadow rays of a generation and the child rays of the next
                                             in two launches instead of one (diagnostic; same results) */
    RT_FLAG_PER_BOUNCE_LAUNCHES = 1u << 3,  /* always trace generation by generation: one trace launch per bounce, shading and
                                               shadow slots between them */
    RT_FLAG_ONE_LAUNCH = 1u << 4,           /* always trace all generations of a chunk in ONE persistent launch in which every
                                               trace workgroup shades the rays it finishes and traces what they spawn.
                                               With neither flag the library chooses by size: one launch for renders of up to
                                               2^20 primary rays (fewer launches, no per-generation end games), per bounce
                                               above (higher throughput).  Same rays, same results, byte for byte, either way.
                                               Ignored where shading draws random numbers and by the _noalloc entry point. */
    RT_FLAG_NEAR_FIRST = 1u << 2          /* NON-PARITY mode (off by default): closest-hit rays visit the children of a
                                             BVH node nearest first instead of in the fixed left-then-right order of
                                             IntersectTopBVH / IntersectMeshBVH (raytracer.cpp:749-753, 802-806).
                                             The closest hit is still the minimum over the same triangles, so frames
                                             differ from the reference's only where two candidates tie or where the
                                             reference's result depends on the order it met transformed instances
                                             in (raytracer.cpp:722-733); tests/test_gpu_parity.py holds the measured
                                             bound (hit ids >= 99.99 %, colours within 1 LSB on >= 99.9 % of pixels) */
};
enum {
    RT_OPT_FLAGS          = 1,  /* bitmask of RT_FLAG_*                            */
    RT_OPT_POOL_RECORDS   = 2,  /* upper bound of the ray-record pool (records, ~104 B each; default 256 M):
                                   a render allocates four records per sample of its largest chunk, up to this */
    RT_OPT_CHUNK_SAMPLES  = 3,  /* max pixel-samples started per chunk (default 64 M) */
    RT_OPT_STACK_CAP      = 4   /* DIAGNOSTIC: traversal stack entries per ray instead of what the upload derived from the
                                   scene's BVH depths (the derived value is >= 64, so that every BVH the reference can walk
                                   with its stack[64], raytracer.cpp:533,766, fits).  Accepted: 11 .. 4096 (more than the
                                   entries kept in LDS), values below the derived one make deep rays fail with
                                   RT_ERR_OVERFLOW — tests use that.  0 = the derived value again, from the next render on */
};

/* Number of HIP devices visible to this process (<0: rt_status). */
int rt_device_count(void);

/* Copy the flattened scene to `device` and build the device layout (128-B records
 * of four BVH grandchildren, 36-B leaf-ordered triangle records, per-object
 * transform records).
 * Replaces nothing in the reference: it is the set-up that precedes the loop
 * at raytracer.cpp:30.  Synchronous. */
int rt_upload_scene(int device, const rt_scene_desc* scene, rt_scene** out);

/* rt_upload_scene with flags.  RT_UPLOAD_REBUILD_MESH_BVH: NON-PARITY mode — the mesh BVHs of the scene
 * (mesh_nodes / mesh_prim_face, built by SubdivideMesh's midpoint splits, raytracer.cpp:296-339, which leaves deep
 * trees and, where centroids coincide, leaves of thousands of triangles: blog/hw2/episode2.md:240-248) are ignored
 * and rebuilt by the library with a binned surface-area heuristic over the same faces; the scene is then always
 * traced as with RT_FLAG_NEAR_FIRST, ties going to the triangle that comes first in the CALLER's leaf order.
 * Frames equal the reference's wherever its box culls agree with its triangle tests; hit ids and
 * colours are tested to the north_star's tolerance, not to byte-exactness.  flags == 0 is rt_upload_scene.
 *
 * RT_UPLOAD_REBUILD_ON_DEVICE (with RT_UPLOAD_REBUILD_MESH_BVH): the rebuild runs on the GPU — the same binned SAH,
 * decision for decision, level by level over the whole mesh and one wave per subtree of 64 triangles (rt_dsah.h): the
 * host builder's trees in a few milliseconds of kernels instead of ~0.15 s of host threads per million triangles.
 * RT_UPLOAD_REBUILD_LBVH (with both): a linear BVH instead (Morton sort + Karras' radix tree, rt_lbvh.h) — the
 * cheapest build, for trees that trace 5-15 % slower than the SAH ones.  Same tolerance, same tie rule. */
enum { RT_UPLOAD_REBUILD_MESH_BVH = 1u, RT_UPLOAD_REBUILD_ON_DEVICE = 2u, RT_UPLOAD_REBUILD_LBVH = 4u };
int rt_upload_scene_ex(int device, const rt_scene_desc* scene, uint32_t flags, rt_scene** out);

/* Next frame of an animation whose meshes do not change: re-derive everything
 * in `scene` except the mesh level (mesh_bvhs, mesh_nodes, mesh_prim_face,
 * faces, vertex_data must be what rt_upload_scene saw) and keep the mesh BVHs,
 * triangle records and shading data resident on the device.  What may change:
 * object transforms / motion blur / materials (meshes, spheres, triangles),
 * the top-level BVH and its prims, planes, materials, point and area lights,
 * background, ambient, epsilon, recursion depth (the camera is an argument of
 * the render calls anyway).  The reference renders an animation as one process
 * per frame (render_and_make_gif.sh:102-119: parse, BuildAllMeshBVHs,
 * BuildTopLevelBVH, render — raytracer.cpp:20-33 — for every JSON); with this
 * call only the per-frame part of that set-up is repeated.
 * flags: RT_UPDATE_VERIFY_MESHES compares a 64-bit hash of the mesh arrays with
 * the one recorded at upload (array sizes are always compared) and fails with
 * RT_ERR_INVALID on a difference.  Synchronous; no render call of this scene
 * may be in flight. */
enum { RT_UPDATE_VERIFY_MESHES = 1u };
int rt_update_scene(rt_scene* s, const rt_scene_desc* scene, uint32_t flags);

int rt_scene_set_option(rt_scene* s, int option, uint64_t value);

/* Render the pixel rectangle [x0,x0+tile_w) x [y0,y0+tile_h) of camera `cam`
 * into caller-owned HOST memory.  Replaces the loop body raytracer.cpp:42-86
 * for those pixels.  rgb_out: row-major RGB8, row 0 = top row of the tile,
 * rgb_row_stride_bytes between rows (raytracer.cpp:82-85 layout).
 * hit_id_out (nullable): 2 x u32 per pixel, row-major, tile_w*2 words per row:
 *   word0 = kind<<28 | object index, word1 = faceIdx inside the mesh (0 for
 *   other kinds); 0xFFFFFFFF,0xFFFFFFFF = miss.  It is the primary hit of the
 *   pixel's FIRST sample — the reference never exposes it; it is what
 *   IntersectTopBVH returns in closestType/index/hitTriFace
 *   (raytracer.cpp:527-531).
 * Synchronous on return. */
int rt_render_tile(rt_scene* s, const rt_camera_desc* cam,
                   int x0, int y0, int tile_w, int tile_h,
                   uint8_t* rgb_out, size_t rgb_row_stride_bytes,
                   uint32_t* hit_id_out, rt_stats* stats);

/* Same, but rgb_out / hit_id_out are DEVICE pointers on the scene's GPU and the
 * work is enqueued on `hip_stream` (a hipStream_t passed as void*; NULL = the
 * scene's own stream).  Returns after enqueueing unless stats != NULL or
 * RT_FLAG_TIMING is set (or the scene spawns child rays / draws random numbers
 * while shading), in which case it synchronises the stream to fill stats.
 * When it returns early, a failure the kernels detect (RT_ERR_OVERFLOW, an
 * internal bound) is not lost: it is returned — once — by rt_sync() or by
 * the next render / update call on this scene, and the frame that render
 * wrote is then not to be used.  Used for multi-GPU frames: every GPU renders its tiles into its own
 * device framebuffer and only that framebuffer is gathered (xGMI). */
int rt_render_tile_device(rt_scene* s, const rt_camera_desc* cam,
                          int x0, int y0, int tile_w, int tile_h,
                          void* d_rgb_out, size_t rgb_row_stride_bytes,
                          void* d_hit_id_out, rt_stats* stats, void* hip_stream);

/* rt_render_tile_device for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc,
 * HIP-graph capture (see the end of this comment for what can be captured).  It has no counterpart in the reference (whose loop allocates nothing but its image,
 * raytracer.cpp:34).  The call allocates NOTHING — no device or pinned memory, no events: everything a render
 * otherwise keeps in buffers of the scene (control block, traversal spill stacks and donation candidates, the
 * ray-record pool and shadow slots of a chunk, the MT19937 state of multi-sample chunks) lives in one 32 MiB
 * array inside the library's code object, of which every GPU the library runs on holds its own copy until the
 * process ends.  The trace kernels run on a reduced grid (16 workgroups), so this is the entry point for small
 * tiles, not for frames.  Limits: tile_w * tile_h * num_samples <= 4 194 304 (RT_ERR_INVALID beyond); scenes
 * whose shading draws random numbers (area lights, Roughness > 0) are refused (RT_ERR_UNSUPPORTED), as is a
 * scene whose BVHs are so deep that the spill stacks do not fit.  Output, hit ids and chunking behave as in
 * rt_render_tile_device (no stats); it returns after enqueueing when the scene spawns no child rays and
 * synchronises the stream otherwise (a chunk's pool-overflow check), failures of an early return being reported
 * as there.  All calls on one GPU share that GPU's array, whatever the scene: calls that overlap in time on one
 * GPU (other streams, other threads, other scenes) race — serialise them.
 * Stream capture: a call on a scene WITHOUT reflective materials (no child rays: it only enqueues) may be recorded
 * with hipStreamBeginCapture on `hip_stream` — call rt_sync() first, so that no earlier render's check is pending —
 * and replayed with hipGraphLaunch; after synchronising the stream of a replay, rt_sync() reports what that replay's
 * kernels left (overflow, internal bounds).  A scene with reflective materials synchronises the stream after every
 * chunk (the pool-overflow check) and cannot be captured; tests/test_zz_no_alloc.py records and replays a graph. */
int rt_render_tile_device_noalloc(rt_scene* s, const rt_camera_desc* cam,
                                  int x0, int y0, int tile_w, int tile_h,
                                  void* d_rgb_out, size_t rgb_row_stride_bytes,
                                  void* d_hit_id_out, void* hip_stream);

/* Multi-GPU load balancing: as rt_render_tile_device, but only the 8-pixel-high
 * tile rows ty (counted from the top of the rectangle) with ty % row_parts ==
 * row_part are rendered; all other bytes of the output buffers are left
 * untouched.  GPU g of G calls it with (G, g) on its own replica of the scene
 * and its own device framebuffer; the framebuffers are then merged (RCCL
 * gather / peer copy of each GPU's rows).  Pixels are independent and seeded by
 * their global coordinates (raytracer.cpp:44), so the merged image is
 * byte-identical to a single-GPU render.  The reference's counterpart is the
 * static OpenMP partition of the same loop (raytracer.cpp:40). */
int rt_render_tile_rows_device(rt_scene* s, const rt_camera_desc* cam,
                               int x0, int y0, int tile_w, int tile_h,
                               int row_parts, int row_part,
                               void* d_rgb_out, size_t rgb_row_stride_bytes,
                               void* d_hit_id_out, rt_stats* stats, void* hip_stream);

/* Frame assembly of a multi-GPU render: what follows n calls of rt_render_tile_rows_device(per_gpu[g], cam, x0, y0,
 * width, height, n, g, d_fb[g], row_stride_bytes, ...) on n replicas of the scene.  d_fb[g] is GPU g's device
 * frame of the rectangle (height rows of row_stride_bytes, of which it wrote its 8-row bands ty = g, g + n, ...).
 * GPU g != 0 copies its bands into d_fb[0] — one strided device-to-device copy over its direct xGMI link to GPU 0,
 * enqueued on hip_streams[g] (NULL array or entry: the scene's own stream), i.e. behind the render enqueued there —
 * and GPU 0's stream waits for all of them.  host_rgb != NULL: the assembled rectangle is then copied to host_rgb
 * (height rows of width * 3 bytes, tightly packed; pinned memory preferred) and the call returns when it is there.
 * host_rgb == NULL: the call returns after enqueueing; d_fb[0] holds the frame for work enqueued later on GPU 0's
 * stream.  Only RGB8 rows cross the links (north_star: "only the final framebuffer is gathered"); no collective
 * library is involved — the processes-per-GPU form of the same gather is raytracer_amd/multigpu.py (RCCL).
 * The reference's counterpart: the one contiguous w*h*3 image its OpenMP team writes (raytracer.cpp:34,82-85) and
 * stbi_write_png consumes (raytracer.cpp:89).  All scenes may live on one device (rehearsal on a 1-GPU box).
 * Allocates nothing (events are made at upload). */
int rt_gather_rows(rt_scene* const* per_gpu, int n, void* const* d_fb, size_t row_stride_bytes,
                   int width, int height, void* const* hip_streams, uint8_t* host_rgb);

/* Debug aid: copy `count` entries of one array of the ray-record pool, as left
 * by the last render call, to host memory.  which: 0 ray0 (origin.xyz,time; 16 B),
 * 1 ray1 (direction.xyz,live; 16 B), 2 hit0 (t,kind_index,tri,-; 16 B),
 * 3 normal+material (16 B), 4 link (child0,child1,slot of light 0,flags | K<<8:
 * the record's slot of light l is slot0 + l*K; 16 B),
 * 5 colour (16 B), 6 shadow_lit (1 B per shadow slot).  With one chunk and
 * num_samples == 1, record r is pixel (8*(T%tiles_per_row)+(r&7), 8*(T/tiles_per_row)+((r&63)>>3)),
 * T = r>>6, of the rendered rectangle.  No counterpart in the reference. */
int rt_debug_fetch(rt_scene* s, int which, uint64_t first, uint64_t count, void* out);

/* Wait for everything enqueued on the scene's stream and for the last render
 * that returned before its kernels ran (also one enqueued on a caller's
 * stream); returns that render's failure, if any (see rt_render_tile_device). */
int rt_sync(rt_scene* s);

void rt_free_scene(rt_scene* s);

const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_H */
