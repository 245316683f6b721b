/*
 * rt_host.h — C ABI of the host-side helper library (librt_host.so).
 *
 * Pure host code, no GPU and no reference dependency.  It owns flattened scenes
 * (the rt_scene_desc of rt.h plus cameras), reads/writes them as RTSCENE1 blobs,
 * and can BUILD scenes programmatically for synthetic workloads: it then does,
 * on the host, what the reference's parser and BVH builders do for a JSON file
 * (face normals parser.cpp:609-637, bounds parser.cpp:113-135, midpoint-split
 * BVHs raytracer.cpp:98-339, Gauss-Jordan inverse Mat4f.h:61-123, camera frame
 * parser.cpp:343-446) so that a scene built here and the same scene written as
 * JSON+PLY and parsed by the reference flatten to the same arrays.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include "rt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

rth_scene*            rth_scene_load(const char* path);               /* NULL on failure */
int                   rth_scene_save(const rth_scene* s, const char* path);
const rt_scene_desc*  rth_scene_desc(const rth_scene* s);
int                   rth_scene_num_cameras(const rth_scene* s);
const rt_camera_desc* rth_scene_camera(const rth_scene* s, int i);
const char*           rth_scene_camera_name(const rth_scene* s, int i);
/* replace camera i's resolution / sample count, recomputing pixel size and the
 * stratification side exactly as parser.cpp:405-443 does (q is unchanged). */
int                   rth_scene_set_camera_resolution(rth_scene* s, int i, int width, int height, int num_samples);
/* override Scene::max_recursion_depth (tests sweep it) */
int                   rth_scene_set_max_depth(rth_scene* s, int max_recursion_depth);
void                  rth_scene_free(rth_scene* s);
const char*           rth_last_error(void);

/* ---- programmatic scene construction ---- */
typedef struct rth_builder rth_builder;
rth_builder* rth_builder_new(void);
void         rth_builder_free(rth_builder* b);
void rth_builder_set_globals(rth_builder* b, const float background[3], const float ambient[3],
                             float shadow_ray_epsilon, int max_recursion_depth);
/* returns the 1-based id of the first vertex added */
int  rth_builder_add_vertices(rth_builder* b, const float* xyz, int n);
int  rth_builder_add_material(rth_builder* b, const rt_material* m);          /* returns 1-based id */
void rth_builder_add_point_light(rth_builder* b, const float pos[3], const float intensity[3]);
void rth_builder_add_area_light(rth_builder* b, const float pos[3], const float normal[3], float size,
                                const float radiance[3]);
/* faces: 3*n 1-based vertex ids.  transform: 16 floats row-major or NULL.
 * motion_blur: 3 floats or NULL.  Returns the mesh index (position in `meshes`). */
int  rth_builder_add_mesh(rth_builder* b, const int32_t* faces, int n_faces, int material_id, int smooth,
                          const float* transform, const float* motion_blur);
/* MeshInstance (parser.cpp:680-747): final = transform * base.transform unless reset. */
int  rth_builder_add_instance(rth_builder* b, int base_mesh_index, int material_id, const float* transform,
                              int reset_transform, const float* motion_blur);
int  rth_builder_add_triangle(rth_builder* b, int i0, int i1, int i2, int material_id, const float* transform,
                              const float* motion_blur);
int  rth_builder_add_sphere(rth_builder* b, int center_vertex_id, float radius, int material_id,
                            const float* transform, const float* motion_blur);
int  rth_builder_add_plane(rth_builder* b, int point_vertex_id, const float normal[3], int material_id);
/* lookAt camera (parser.cpp:343-446): fovy in degrees. aperture_size <= 0: pinhole. */
int  rth_builder_add_camera_lookat(rth_builder* b, const float position[3], const float gaze_point[3],
                                   const float up[3], float fovy_deg, float near_distance, int width, int height,
                                   int num_samples, float aperture_size, float focus_distance, const char* name);
/* Builds all BVHs and returns the finished scene (builder stays valid). */
rth_scene* rth_builder_finish(rth_builder* b);

/* Affine helpers matching parser.cpp:32-70 (row-major 4x4 out). */
void rth_mat_translation(float out[16], float x, float y, float z);
void rth_mat_scaling(float out[16], float x, float y, float z);
void rth_mat_rotation(float out[16], float angle_deg, float ax, float ay, float az);
void rth_mat_mul(float out[16], const float a[16], const float b[16]);   /* out = a * b */

#ifdef __cplusplus
}
#endif
#endif
