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


def hit_ids_match(hits_u32, ref6, scene_faces=None):
    """Fraction of pixels whose (kind, index) equal the reference's.  hits_u32: [h,w,2] as rt.h documents."""
    miss = hits_u32[..., 0] == 0xFFFFFFFF
    kind = np.where(miss, -1, (hits_u32[..., 0] >> 28).astype(np.int64))
    index = (hits_u32[..., 0] & 0x0FFFFFFF).astype(np.int64)
    ok = (kind == ref6[..., 0]) & (miss | (index == ref6[..., 1]))
    return ok


def image_stats(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=-1)
    return {"exact": float((d == 0).mean()), "within1": float((d <= 1).mean()), "maxdiff": int(d.max())}
