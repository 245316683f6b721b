"""ctypes bindings for oracle/librt_oracle.so (the C restatement) and, when it
has been built, oracle/_ref/libref_oracle.so (the reference itself)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Counters(C.Structure):
    """rto_counters (oracle/rt_oracle.c)"""
    _fields_ = [(n, C.c_uint64) for n in (
        "c_rays", "c_top_nodes", "c_mesh_nodes", "c_tri_tests", "c_sphere_tests", "c_plane_tests", "c_inst_entries",
        "s_rays", "s_top_nodes", "s_mesh_nodes", "s_tri_tests", "s_sphere_tests", "s_plane_tests", "s_inst_entries",
        "rays_primary", "rays_secondary", "max_stack")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    def closest_bytes(self):
        """Algorithmic bytes of the closest-hit rays, SURVEY.md §8(d):
        32*(N_top+N_mesh) + 36*T_tri + 16*T_sph + 16*T_pln + 96*E_inst + 48 per ray."""
        return (32 * (self.c_top_nodes + self.c_mesh_nodes) + 36 * self.c_tri_tests + 16 * self.c_sphere_tests +
                16 * self.c_plane_tests + 96 * self.c_inst_entries + 48 * self.c_rays)

    def shadow_bytes(self):
        return (32 * (self.s_top_nodes + self.s_mesh_nodes) + 36 * self.s_tri_tests + 16 * self.s_sphere_tests +
                16 * self.s_plane_tests + 96 * self.s_inst_entries + 48 * self.s_rays)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(_HERE, "librt_oracle.so"))
        L.rto_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                 C.c_void_p, C.POINTER(Counters), C.c_int]
        L.rto_rng_kat.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def render(host_scene, cam, x0=0, y0=0, w=None, h=None, want_rgb=True, want_hits=False, threads=0):
    """Run the restatement.  Returns (rgb or None, hits or None, Counters)."""
    w = cam.image_width if w is None else w
    h = cam.image_height if h is None else h
    rgb = np.zeros((h, w, 3), np.uint8) if want_rgb else None
    hits = np.zeros((h, w, 2), np.uint32) if want_hits else None
    ctr = Counters()
    rc = lib().rto_render(host_scene.desc_ptr, C.byref(cam), x0, y0, w, h,
                          rgb.ctypes.data if want_rgb else None, w * 3,
                          hits.ctypes.data if want_hits else None, C.byref(ctr), threads)
    if rc != 0:
        raise RuntimeError(f"rto_render failed: {rc}")
    return rgb, hits, ctr


def rng_kat(seed, n_u32, n_perm, n_f):
    raw = np.zeros(n_u32, np.uint32)
    perm = np.zeros(max(n_perm, 1), np.int32)
    f = np.zeros(max(n_f, 1), np.float32)
    lib().rto_rng_kat(seed, raw.ctypes.data, n_u32, perm.ctypes.data, n_perm, f.ctypes.data, n_f)
    return raw, perm[:n_perm], f[:n_f]


# ---------------------------------------------------------------- reference
REF_SO = os.path.join(_HERE, "_ref", "libref_oracle.so")
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The reference itself (compiled from /root/reference by oracle/Makefile)."""
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_load.argtypes = [C.c_char_p]
        L.ref_camera.argtypes = [C.c_int, C.c_void_p]
        L.ref_save_flat.argtypes = [C.c_char_p]
        L.ref_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ref_primary_hits.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_last_render_seconds.restype = C.c_double
        _ref = L
    return _ref


def ref_load(json_path):
    if ref().ref_load(os.fsencode(json_path)) != 0:
        raise RuntimeError(f"reference failed to load {json_path}")


def ref_render(cam_index, x0, y0, w, h, threads=0):
    rgb = np.zeros((h, w, 3), np.uint8)
    if ref().ref_render(cam_index, x0, y0, w, h, rgb.ctypes.data, threads) != 0:
        raise RuntimeError("ref_render failed")
    return rgb, ref().ref_last_render_seconds()


def ref_primary_hits(cam_index, x0, y0, w, h):
    out = np.zeros((h, w, 6), np.int32)
    if ref().ref_primary_hits(cam_index, x0, y0, w, h, out.ctypes.data) != 0:
        raise RuntimeError("ref_primary_hits failed")
    return out
