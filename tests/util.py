"""Shared helpers for the test-suite (fixtures under tests/golden/)."""
import gzip
import os
import shutil
import tempfile

import numpy as np
from PIL import Image

from raytracer_amd import HostScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden_scene(name):
    """HostScene of tests/golden/<name>.rtscene.gz (flattened by the reference's own parser/BVH build)."""
    if name not in _cache:
        with tempfile.NamedTemporaryFile(suffix=".rtscene", delete=False) as tmp:
            with gzip.open(os.path.join(GOLDEN, name + ".rtscene.gz"), "rb") as g:
                shutil.copyfileobj(g, tmp)
            path = tmp.name
        try:
            _cache[name] = HostScene.load(path)
        finally:
            os.remove(path)
    return _cache[name]


def golden_image(name):
    return np.array(Image.open(os.path.join(GOLDEN, name + ".png")).convert("RGB"))


def golden_hits(name):
    """int32 [h,w,6]: kind (-1 miss), index, i0, i1, i2, t bits — from the reference's IntersectTopBVH."""
    return np.load(os.path.join(GOLDEN, name + ".hits.npz"))["hits"]


def golden_tile(name):
    """(x0, y0, w, h) of the rectangle the fixture's png and hits cover, or None for the whole frame."""
    z = np.load(os.path.join(GOLDEN, name + ".hits.npz"))
    return tuple(int(v) for v in z["tile"]) if "tile" in z.files else None


def scene_faces(hs):
    """Faces / mesh tables read back out of the rt_scene_desc: vertex ids per face, face_offset per mesh BVH,
    mesh BVH per scene.meshes entry."""
    import ctypes as C
    from raytracer_amd.ffi import SceneDesc
    d = SceneDesc.from_address(hs.desc_ptr)
    faces = np.ctypeslib.as_array((C.c_int32 * (7 * d.n_faces)).from_address(d.faces)).reshape(-1, 7)[:, :3].copy() if d.n_faces else np.zeros((0, 3), np.int32)
    bv = np.array([[d.mesh_bvhs[b].face_offset] for b in range(d.n_mesh_bvhs)], np.int64).reshape(-1)
    mesh_words = 3 + 16 + 16 + 3          # rt_mesh: bvh_index, material_id, flags, two matrices, motion blur
    ms = np.ctypeslib.as_array((C.c_int32 * (mesh_words * d.n_meshes)).from_address(d.meshes)).reshape(-1, mesh_words).copy() if d.n_meshes else np.zeros((0, mesh_words), np.int32)
    return {"tri": faces, "face_offset": bv, "mesh_bvh": ms[:, 0].astype(np.int64)}


def hit_ids_match(hits_u32, ref6, hs=None):
    """Per pixel: is the primary hit the reference's?  hits_u32: [h,w,2] as rt.h documents (kind<<28 | object
    index, faceIdx); ref6: the fixture (kind, index, i0, i1, i2, t bits from the reference's IntersectTopBVH,
    raytracer.cpp:527-531).  (kind, index) must be equal; with `hs` (the scene) a mesh hit's faceIdx is also
    joined to its vertex ids and must name the triangle (i0, i1, i2) the reference returned."""
    miss = hits_u32[..., 0] == 0xFFFFFFFF
    kind = np.where(miss, -1, (hits_u32[..., 0] >> 28).astype(np.int64))
    index = (hits_u32[..., 0] & 0x0FFFFFFF).astype(np.int64)
    ok = (kind == ref6[..., 0]) & (miss | (index == ref6[..., 1]))
    if hs is not None:
        m = ok & (kind == 0)
        if m.any():
            f = scene_faces(hs)
            fi = hits_u32[..., 1][m].astype(np.int64)
            tri = f["tri"][f["face_offset"][f["mesh_bvh"][index[m]]] + fi]
            ok[m] = (tri == ref6[..., 2:5][m]).all(axis=-1)
    return ok


def image_stats(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=-1)
    return {"exact": float((d == 0).mean()), "within1": float((d <= 1).mean()), "maxdiff": int(d.max())}
