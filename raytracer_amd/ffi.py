"""ctypes bindings for the C ABI in include/rt.h and include/rt_host.h.

Python is plumbing here (tests, bench, multi-GPU launch); the product is
librt_hip.so.  There is no CPU fallback: if librt_hip.so is missing or no HIP
device is present, every render entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

RT_KIND_MESH, RT_KIND_TRIANGLE, RT_KIND_SPHERE, RT_KIND_PLANE = 0, 1, 2, 3
RT_MAT_NONE, RT_MAT_MIRROR, RT_MAT_CONDUCTOR, RT_MAT_DIELECTRIC = 0, 1, 2, 3
RT_FLAG_TIMING = 1
RT_FLAG_SEPARATE_LAUNCHES = 2
RT_FLAG_NEAR_FIRST = 4        # non-parity visiting order, see include/rt.h
RT_FLAG_PER_BOUNCE_LAUNCHES = 8
RT_FLAG_ONE_LAUNCH = 16
RT_UPLOAD_REBUILD_MESH_BVH = 1
RT_UPLOAD_REBUILD_ON_DEVICE = 2
RT_UPLOAD_REBUILD_LBVH = 4
RT_UPDATE_VERIFY_MESHES = 1
RT_OPT_FLAGS, RT_OPT_POOL_RECORDS, RT_OPT_CHUNK_SAMPLES, RT_OPT_STACK_CAP = 1, 2, 3, 4


class RtError(RuntimeError):
    pass


class CameraDesc(C.Structure):
    """rt_camera_desc (include/rt.h)"""
    _fields_ = [
        ("position", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
        ("q", C.c_float * 3), ("pixel_width", C.c_float), ("pixel_height", C.c_float),
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("num_samples", C.c_int32), ("samples_per_side", C.c_int32),
        ("aperture_size", C.c_float), ("focus_distance", C.c_float), ("has_depth_of_field", C.c_int32),
    ]


class Material(C.Structure):
    """rt_material (include/rt.h)"""
    _fields_ = [
        ("type", C.c_uint32), ("ambient_refl", C.c_float * 3), ("diffuse_refl", C.c_float * 3),
        ("specular_refl", C.c_float * 3), ("mirror_refl", C.c_float * 3), ("phong_exponent", C.c_float),
        ("refraction_index", C.c_float), ("absorption_index", C.c_float), ("absorption_coef", C.c_float * 3),
        ("roughness", C.c_float),
    ]


class Stats(C.Structure):
    """rt_stats (include/rt.h)"""
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("records_peak", C.c_uint64), ("chunks", C.c_uint32), ("launches_closest", C.c_uint32),
        ("launches_shadow", C.c_uint32), ("ms_total", C.c_float), ("ms_raygen", C.c_float),
        ("ms_closest", C.c_float), ("ms_shade", C.c_float), ("ms_shadow", C.c_float),
        ("ms_combine", C.c_float), ("ms_resolve", C.c_float),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BvhNode(C.Structure):
    """rt_bvh_node (include/rt.h)"""
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left_or_first", C.c_int32), ("prim_count", C.c_int32)]


class MeshBvh(C.Structure):
    """rt_mesh_bvh (include/rt.h)"""
    _fields_ = [("node_offset", C.c_uint32), ("node_count", C.c_uint32), ("prim_offset", C.c_uint32),
                ("prim_count", C.c_uint32), ("face_offset", C.c_uint32), ("face_count", C.c_uint32)]


class SceneDesc(C.Structure):
    """rt_scene_desc (include/rt.h): a view of the pointers + counts, for tests that inspect or alter a scene."""
    _fields_ = [
        ("background_color", C.c_float * 3), ("ambient_light", C.c_float * 3), ("shadow_ray_epsilon", C.c_float),
        ("max_recursion_depth", C.c_int32),
        ("top_nodes", C.POINTER(BvhNode)), ("n_top_nodes", C.c_uint32),
        ("top_prim_idx", C.POINTER(C.c_uint32)), ("n_top_prims", C.c_uint32),
        ("top_prims", C.c_void_p),
        ("mesh_bvhs", C.POINTER(MeshBvh)), ("n_mesh_bvhs", C.c_uint32),
        ("mesh_nodes", C.POINTER(BvhNode)), ("n_mesh_nodes", C.c_uint32),
        ("mesh_prim_face", C.POINTER(C.c_uint32)), ("n_mesh_prims", C.c_uint32),
        ("faces", C.c_void_p), ("n_faces", C.c_uint32),
        ("vertex_data", C.c_void_p), ("n_vertices", C.c_uint32),
        ("meshes", C.c_void_p), ("n_meshes", C.c_uint32),
        ("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
        ("spheres", C.c_void_p), ("n_spheres", C.c_uint32),
        ("planes", C.c_void_p), ("n_planes", C.c_uint32),
        ("materials", C.c_void_p), ("n_materials", C.c_uint32),
        ("point_lights", C.c_void_p), ("n_point_lights", C.c_uint32),
        ("area_lights", C.c_void_p), ("n_area_lights", C.c_uint32),
    ]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _mat16(m):
    if m is None:
        return None
    flat = [float(x) for row in m for x in (row if hasattr(row, "__len__") else [row])]
    assert len(flat) == 16
    return (C.c_float * 16)(*flat)


# ------------------------------------------------------------------ host lib
def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RtError(f"{name} is not built (run `make` or __graft_entry__.build())")
    return C.CDLL(path)


_host = None


def host_lib():
    global _host
    if _host is None:
        L = _load("librt_host.so")
        L.rth_scene_load.restype = C.c_void_p
        L.rth_scene_load.argtypes = [C.c_char_p]
        L.rth_scene_save.argtypes = [C.c_void_p, C.c_char_p]
        L.rth_scene_desc.restype = C.c_void_p
        L.rth_scene_desc.argtypes = [C.c_void_p]
        L.rth_scene_num_cameras.argtypes = [C.c_void_p]
        L.rth_scene_camera.restype = C.POINTER(CameraDesc)
        L.rth_scene_camera.argtypes = [C.c_void_p, C.c_int]
        L.rth_scene_camera_name.restype = C.c_char_p
        L.rth_scene_camera_name.argtypes = [C.c_void_p, C.c_int]
        L.rth_scene_set_camera_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_scene_set_max_depth.argtypes = [C.c_void_p, C.c_int]
        L.rth_last_error.restype = C.c_char_p
        L.rth_builder_new.restype = C.c_void_p
        L.rth_builder_free.argtypes = [C.c_void_p]
        L.rth_builder_set_globals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int]
        L.rth_builder_add_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rth_builder_add_material.argtypes = [C.c_void_p, C.POINTER(Material)]
        L.rth_builder_add_point_light.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_builder_add_area_light.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.rth_builder_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rth_builder_add_instance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rth_builder_add_triangle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rth_builder_add_sphere.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.rth_builder_add_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.rth_builder_add_camera_lookat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                                    C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                    C.c_char_p]
        L.rth_builder_finish.restype = C.c_void_p
        L.rth_builder_finish.argtypes = [C.c_void_p]
        for fn in ("rth_mat_translation", "rth_mat_scaling"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.rth_mat_rotation.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.rth_mat_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _host = L
    return _host


class HostScene:
    """A flattened scene (rt_scene_desc + cameras) living in host memory."""

    def __init__(self, handle):
        if not handle:
            raise RtError(host_lib().rth_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def load(cls, path):
        return cls(host_lib().rth_scene_load(os.fsencode(path)))

    def save(self, path):
        if host_lib().rth_scene_save(self._h, os.fsencode(path)) != 0:
            raise RtError(host_lib().rth_last_error().decode())

    @property
    def desc_ptr(self):
        return host_lib().rth_scene_desc(self._h)

    @property
    def num_cameras(self):
        return host_lib().rth_scene_num_cameras(self._h)

    def camera(self, i=0):
        p = host_lib().rth_scene_camera(self._h, i)
        if not p:
            raise IndexError(i)
        c = CameraDesc()
        C.memmove(C.byref(c), p, C.sizeof(CameraDesc))
        return c

    def camera_name(self, i=0):
        return host_lib().rth_scene_camera_name(self._h, i).decode()

    def set_camera_resolution(self, i, width, height, num_samples):
        if host_lib().rth_scene_set_camera_resolution(self._h, i, width, height, num_samples) != 0:
            raise RtError("bad camera resolution")

    def set_max_depth(self, depth):
        host_lib().rth_scene_set_max_depth(self._h, depth)

    def close(self):
        if self._h:
            host_lib().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mat_translation(x, y, z):
    m = (C.c_float * 16)()
    host_lib().rth_mat_translation(m, x, y, z)
    return m


def mat_scaling(x, y, z):
    m = (C.c_float * 16)()
    host_lib().rth_mat_scaling(m, x, y, z)
    return m


def mat_rotation(angle_deg, ax, ay, az):
    m = (C.c_float * 16)()
    host_lib().rth_mat_rotation(m, angle_deg, ax, ay, az)
    return m


def mat_mul(a, b):
    m = (C.c_float * 16)()
    host_lib().rth_mat_mul(m, a, b)
    return m


class SceneBuilder:
    """Programmatic scene construction (rth_builder_*); mirrors what the
    reference's JSON parser accepts (parser.cpp:139-930)."""

    def __init__(self):
        self._L = host_lib()
        self._b = C.c_void_p(self._L.rth_builder_new())

    def set_globals(self, background=(0, 0, 0), ambient=(0, 0, 0), shadow_ray_epsilon=1e-3, max_recursion_depth=6):
        self._L.rth_builder_set_globals(self._b, _f3(background), _f3(ambient), shadow_ray_epsilon, max_recursion_depth)

    def add_vertices(self, xyz):
        import numpy as np
        a = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        return self._L.rth_builder_add_vertices(self._b, a.ctypes.data, a.shape[0])

    def add_material(self, type=RT_MAT_NONE, ambient=(0, 0, 0), diffuse=(0, 0, 0), specular=(0, 0, 0),
                     mirror=(0, 0, 0), phong=1.0, refraction_index=0.0, absorption_index=0.0,
                     absorption_coef=(0, 0, 0), roughness=0.0):
        m = Material(type, _f3(ambient), _f3(diffuse), _f3(specular), _f3(mirror), phong, refraction_index,
                     absorption_index, _f3(absorption_coef), roughness)
        return self._L.rth_builder_add_material(self._b, C.byref(m))

    def add_point_light(self, position, intensity):
        self._L.rth_builder_add_point_light(self._b, _f3(position), _f3(intensity))

    def add_area_light(self, position, normal, size, radiance):
        self._L.rth_builder_add_area_light(self._b, _f3(position), _f3(normal), size, _f3(radiance))

    def add_mesh(self, faces, material_id, smooth=False, transform=None, motion_blur=None):
        import numpy as np
        f = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        r = self._L.rth_builder_add_mesh(self._b, f.ctypes.data, f.shape[0], material_id, int(smooth), transform,
                                         _f3(motion_blur) if motion_blur is not None else None)
        if r < 0:
            raise RtError(self._L.rth_last_error().decode())
        return r

    def add_instance(self, base_mesh, material_id=0, transform=None, reset_transform=False, motion_blur=None):
        r = self._L.rth_builder_add_instance(self._b, base_mesh, material_id, transform, int(reset_transform),
                                             _f3(motion_blur) if motion_blur is not None else None)
        if r < 0:
            raise RtError(self._L.rth_last_error().decode())
        return r

    def add_triangle(self, i0, i1, i2, material_id, transform=None, motion_blur=None):
        r = self._L.rth_builder_add_triangle(self._b, i0, i1, i2, material_id, transform,
                                             _f3(motion_blur) if motion_blur is not None else None)
        if r < 0:
            raise RtError(self._L.rth_last_error().decode())
        return r

    def add_sphere(self, center_vertex_id, radius, material_id, transform=None, motion_blur=None):
        r = self._L.rth_builder_add_sphere(self._b, center_vertex_id, radius, material_id, transform,
                                           _f3(motion_blur) if motion_blur is not None else None)
        if r < 0:
            raise RtError(self._L.rth_last_error().decode())
        return r

    def add_plane(self, point_vertex_id, normal, material_id):
        r = self._L.rth_builder_add_plane(self._b, point_vertex_id, _f3(normal), material_id)
        if r < 0:
            raise RtError(self._L.rth_last_error().decode())
        return r

    def add_camera_lookat(self, position, gaze_point, up, fovy_deg, near_distance, width, height, num_samples=1,
                          aperture_size=0.0, focus_distance=1.0, name="out.png"):
        return self._L.rth_builder_add_camera_lookat(self._b, _f3(position), _f3(gaze_point), _f3(up), fovy_deg,
                                                     near_distance, width, height, num_samples, aperture_size,
                                                     focus_distance, name.encode())

    def finish(self):
        return HostScene(self._L.rth_builder_finish(self._b))

    def __del__(self):
        try:
            if self._b:
                self._L.rth_builder_free(self._b)
                self._b = None
        except Exception:
            pass


# ------------------------------------------------------------------- HIP lib
_hip = None


def hip_lib():
    """librt_hip.so — the product.  Raises loudly when it is not built."""
    global _hip
    if _hip is None:
        L = _load(os.environ.get("RT_HIP_SO", "librt_hip.so"))   # RT_HIP_SO: experiment builds (make variant)
        L.rt_device_count.restype = C.c_int
        L.rt_upload_scene.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt_upload_scene_ex.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.rt_scene_set_option.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.rt_update_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.rt_render_tile.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats)]
        L.rt_render_tile_device.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats), C.c_void_p]
        L.rt_gather_rows.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_void_p]
        L.rt_render_tile_rows_device.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Stats),
                                                 C.c_void_p]
        L.rt_render_tile_device_noalloc.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rt_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
        L.rt_sync.argtypes = [C.c_void_p]
        L.rt_free_scene.argtypes = [C.c_void_p]
        L.rt_last_error.restype = C.c_char_p
        _hip = L
    return _hip


def _check(rc):
    if rc != 0:
        raise RtError(f"rt status {rc}: {hip_lib().rt_last_error().decode()}")


class DeviceScene:
    """A scene resident on one GPU (rt_scene*)."""

    def __init__(self, host_scene, device=0, rebuild_mesh_bvh=False):
        """rt_upload_scene; rebuild_mesh_bvh=True is rt_upload_scene_ex(RT_UPLOAD_REBUILD_MESH_BVH), the non-parity
        mode in which the library builds its own SAH trees over the meshes (include/rt.h); rebuild_mesh_bvh="device"
        adds RT_UPLOAD_REBUILD_ON_DEVICE (the same SAH trees built on the GPU), "lbvh" also RT_UPLOAD_REBUILD_LBVH (a
        linear BVH built on the GPU)."""
        L = hip_lib()
        h = C.c_void_p()
        if rebuild_mesh_bvh:
            flags = RT_UPLOAD_REBUILD_MESH_BVH | (RT_UPLOAD_REBUILD_ON_DEVICE if rebuild_mesh_bvh in ("device", "lbvh") else 0) \
                | (RT_UPLOAD_REBUILD_LBVH if rebuild_mesh_bvh == "lbvh" else 0)
            _check(L.rt_upload_scene_ex(device, host_scene.desc_ptr, flags, C.byref(h)))
        else:
            _check(L.rt_upload_scene(device, host_scene.desc_ptr, C.byref(h)))
        self._h = h
        self.device = device

    def set_option(self, option, value):
        _check(hip_lib().rt_scene_set_option(self._h, option, int(value)))

    def update(self, host_scene, verify_meshes=False):
        """rt_update_scene: next animation frame with the same meshes (transforms, lights, materials, top level change)."""
        _check(hip_lib().rt_update_scene(self._h, host_scene.desc_ptr, RT_UPDATE_VERIFY_MESHES if verify_meshes else 0))

    def render(self, cam, x0=0, y0=0, w=None, h=None, want_hits=False, stats=None):
        """rt_render_tile into host numpy arrays.  Returns (rgb[h,w,3], hits[h,w,2] or None)."""
        import numpy as np
        w = cam.image_width if w is None else w
        h = cam.image_height if h is None else h
        rgb = np.zeros((h, w, 3), np.uint8)
        hits = np.zeros((h, w, 2), np.uint32) if want_hits else None
        _check(hip_lib().rt_render_tile(self._h, C.byref(cam), x0, y0, w, h, rgb.ctypes.data, w * 3,
                                        hits.ctypes.data if want_hits else None,
                                        C.byref(stats) if stats is not None else None))
        return rgb, hits

    def render_device(self, cam, x0, y0, w, h, d_rgb_ptr, row_stride, d_hits_ptr=None, stats=None, stream=None):
        """rt_render_tile_device: outputs are device pointers (e.g. torch tensor .data_ptr())."""
        _check(hip_lib().rt_render_tile_device(self._h, C.byref(cam), x0, y0, w, h, C.c_void_p(d_rgb_ptr), row_stride,
                                               C.c_void_p(d_hits_ptr) if d_hits_ptr else None,
                                               C.byref(stats) if stats is not None else None,
                                               C.c_void_p(stream) if stream else None))

    def render_device_noalloc(self, cam, x0, y0, w, h, d_rgb_ptr, row_stride, d_hits_ptr=None, stream=None):
        """rt_render_tile_device_noalloc: the same into device memory without any allocation (static workspace)."""
        _check(hip_lib().rt_render_tile_device_noalloc(self._h, C.byref(cam), x0, y0, w, h, C.c_void_p(d_rgb_ptr), row_stride,
                                                       C.c_void_p(d_hits_ptr) if d_hits_ptr else None,
                                                       C.c_void_p(stream) if stream else None))

    def render_rows_device(self, cam, x0, y0, w, h, row_parts, row_part, d_rgb_ptr, row_stride, d_hits_ptr=None,
                           stats=None, stream=None):
        """rt_render_tile_rows_device: this GPU's interleaved share of the rectangle."""
        _check(hip_lib().rt_render_tile_rows_device(self._h, C.byref(cam), x0, y0, w, h, row_parts, row_part,
                                                    C.c_void_p(d_rgb_ptr), row_stride,
                                                    C.c_void_p(d_hits_ptr) if d_hits_ptr else None,
                                                    C.byref(stats) if stats is not None else None,
                                                    C.c_void_p(stream) if stream else None))

    def sync(self):
        _check(hip_lib().rt_sync(self._h))

    @staticmethod
    def gather_rows(scenes, d_fb_ptrs, row_stride, width, height, host_rgb=None, streams=None):
        """rt_gather_rows: the bands scene g of n rendered into its own device frame d_fb_ptrs[g] are brought together in
        d_fb_ptrs[0] (peer copies) and, with host_rgb (a C-contiguous uint8 numpy array of height x width x 3), on the host."""
        n = len(scenes)
        hs = (C.c_void_p * n)(*[s._h for s in scenes])
        fb = (C.c_void_p * n)(*[C.c_void_p(p) for p in d_fb_ptrs])
        st = (C.c_void_p * n)(*[C.c_void_p(p) for p in streams]) if streams else None
        out = host_rgb.ctypes.data_as(C.c_void_p) if host_rgb is not None else None
        _check(hip_lib().rt_gather_rows(hs, n, fb, row_stride, width, height, st, out))

    def close(self):
        if getattr(self, "_h", None):
            hip_lib().rt_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
