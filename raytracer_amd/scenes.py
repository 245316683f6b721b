"""Synthetic workloads (closed-form geometry, no downloads) for tests and bench.

A scene is described once (SynthScene) and can be realised two ways:
  * .build()      -> HostScene through librt_host's builder (no reference needed);
  * .write_json() -> scene.json + binary-LE PLY files in the reference's own input
                     format (parser.cpp:139-930, 981-1214), so the reference can
                     parse the very same scene (used by tests and by the CPU
                     baseline to time the reference itself).
SURVEY.md §8(d) names the workloads: M2 = `mesh1m` (1 002 528-triangle displaced
sphere, 1920x1080, 2 point lights, depth 0), M3/M5 = `inst10`.
"""
import json
import os

import numpy as np

from . import ffi


def displaced_sphere(n):
    """n x n UV grid on a displaced unit sphere -> (verts float32 [n*n,3], faces int32 [2*(n-1)^2,3], 0-based).
    r = 1 + 0.08 sin(13 theta) sin(11 phi) + 0.03 sin(41 theta + phi)."""
    th = np.linspace(0.0, np.pi, n)[:, None]
    ph = np.linspace(0.0, 2.0 * np.pi, n)[None, :]
    r = 1.0 + 0.08 * np.sin(13 * th) * np.sin(11 * ph) + 0.03 * np.sin(41 * th + ph)
    x = r * np.sin(th) * np.cos(ph)
    y = r * np.cos(th) * np.ones_like(ph)
    z = r * np.sin(th) * np.sin(ph)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).ravel()
    b = a + 1
    c = a + n
    d = c + 1
    faces = np.concatenate([np.stack([a, c, b], -1), np.stack([b, c, d], -1)], 0).astype(np.int32)
    return verts, faces


def _displace(d):
    """Radius of the displaced sphere in direction d (unit vectors [...,3]): the same closed form as
    displaced_sphere, written in terms of the direction so that any tessellation can use it."""
    th = np.arccos(np.clip(d[..., 1], -1.0, 1.0))
    ph = np.arctan2(d[..., 2], d[..., 0]) % (2.0 * np.pi)
    return 1.0 + 0.08 * np.sin(13 * th) * np.sin(11 * ph) + 0.03 * np.sin(41 * th + ph)


def cube_sphere(g):
    """Displaced unit sphere tessellated as a spherified cube: 6 faces of g x g quads -> 12 g^2 triangles of
    near-uniform size (g=289 -> 1 002 252), no poles, no degenerate triangles — the triangle statistics of
    a scanned PLY model rather than of a latitude/longitude grid.  Vertices are indexed by their integer
    lattice position on the cube surface, so faces share edge vertices exactly (watertight).
    Returns (verts float32 [6 g^2 + 2, 3], faces int32 [12 g^2, 3], 0-based)."""
    idx = {}
    pts = []

    def vid(i, j, k):
        key = (i, j, k)
        v = idx.get(key)
        if v is None:
            v = idx[key] = len(pts)
            pts.append(key)
        return v

    faces = []
    r = range(g)
    for axis in range(3):
        for side in (0, g):
            for a in r:
                for b in r:
                    def P(u, v):
                        c = [0, 0, 0]
                        c[axis] = side
                        c[(axis + 1) % 3] = u
                        c[(axis + 2) % 3] = v
                        return vid(*c)
                    q = (P(a, b), P(a + 1, b), P(a + 1, b + 1), P(a, b + 1))
                    if side == 0:
                        q = (q[0], q[3], q[2], q[1])      # outward orientation
                    faces.append((q[0], q[1], q[2]))
                    faces.append((q[0], q[2], q[3]))
    c = np.asarray(pts, np.float64) * (2.0 / g) - 1.0
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    d = np.stack([x * np.sqrt(1 - y * y / 2 - z * z / 2 + y * y * z * z / 3),
                  y * np.sqrt(1 - z * z / 2 - x * x / 2 + z * z * x * x / 3),
                  z * np.sqrt(1 - x * x / 2 - y * y / 2 + x * x * y * y / 3)], -1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    verts = (d * _displace(d)[:, None]).astype(np.float32)
    return verts, np.asarray(faces, np.int32)


def _mat_rows(m):
    return [float(v) for v in m]


class SynthScene:
    def __init__(self, name):
        self.name = name
        self.background = (0, 0, 0)
        self.ambient = (25, 25, 25)
        self.eps = 1e-3
        self.max_depth = 0
        self.vertices = []          # JSON VertexData (list of xyz)
        self.materials = []         # dicts for add_material
        self.point_lights = []      # (pos, intensity)
        self.area_lights = []       # (pos, normal, size, radiance)
        self.meshes = []            # dict(verts|None, faces(0-based, into own verts or 1-based global if verts None), material, smooth, transform)
        self.instances = []         # dict(base (mesh ordinal), material, transform(list of ('t'|'s'|'r', args)), reset)
        self.spheres = []           # dict(center_vertex_id, radius, material, transform)
        self.planes = []            # dict(point_vertex_id, normal, material)
        self.camera = None          # dict

    # ---------------------------------------------------------------- realise
    @staticmethod
    def _compose(ops):
        """'t1 r2 s3' semantics of parser.cpp:72-111: M = last * ... * first."""
        m = ffi.mat_translation(0, 0, 0)
        for kind, args in ops:
            if kind == "t":
                t = ffi.mat_translation(*args)
            elif kind == "s":
                t = ffi.mat_scaling(*args)
            else:
                t = ffi.mat_rotation(*args)
            m = ffi.mat_mul(t, m)
        return m

    def build(self):
        b = ffi.SceneBuilder()
        b.set_globals(self.background, self.ambient, self.eps, self.max_depth)
        if self.vertices:
            b.add_vertices(np.asarray(self.vertices, np.float32))
        for m in self.materials:
            b.add_material(**m)
        for pos, inten in self.point_lights:
            b.add_point_light(pos, inten)
        for pos, nrm, size, rad in self.area_lights:
            b.add_area_light(pos, nrm, size, rad)
        mesh_index = []
        for m in self.meshes:
            if m["verts"] is not None:
                base = b.add_vertices(m["verts"])          # PLY vertices are appended (parser.cpp:571-572)
                faces = m["faces"] + base                   # 1-based global ids
            else:
                faces = m["faces"]
            tr = self._compose(m["transform"]) if m.get("transform") else None
            mesh_index.append(b.add_mesh(faces, m["material"], m.get("smooth", False), tr))
        for inst in self.instances:
            tr = self._compose(inst["transform"]) if inst.get("transform") else None
            mesh_index.append(b.add_instance(mesh_index[inst["base"]], inst.get("material", 0), tr,
                                             inst.get("reset", False)))
        for s in self.spheres:
            tr = self._compose(s["transform"]) if s.get("transform") else None
            b.add_sphere(s["center"], s["radius"], s["material"], tr)
        for p in self.planes:
            b.add_plane(p["point"], p["normal"], p["material"])
        c = self.camera
        b.add_camera_lookat(c["position"], c["gaze_point"], c["up"], c["fovy"], c["near"], c["width"], c["height"],
                            c.get("num_samples", 1), c.get("aperture", 0.0),
                            c.get("focus", 1.0) if c.get("aperture", 0.0) > 0 else 1.0,   # parser.cpp:422 default
                            c.get("name", self.name + ".png"))
        return b.finish()

    # ------------------------------------------------- reference input format
    def write_json(self, out_dir):
        """Write <out_dir>/<name>.json (+ PLYs) in the reference's input format."""
        os.makedirs(out_dir, exist_ok=True)
        fmt = lambda v: " ".join(repr(float(np.float32(x))) for x in v)  # noqa: E731
        tr_tab = {"t": [], "s": [], "r": []}

        def tr_string(ops):
            toks = []
            for kind, args in ops:
                tr_tab[kind].append(args)
                toks.append(f"{kind}{len(tr_tab[kind])}")
            return " ".join(toks)

        mats = []
        names = {ffi.RT_MAT_MIRROR: "mirror", ffi.RT_MAT_CONDUCTOR: "conductor", ffi.RT_MAT_DIELECTRIC: "dielectric"}
        for k, m in enumerate(self.materials):
            d = {"_id": str(k + 1),
                 "AmbientReflectance": fmt(m.get("ambient", (0, 0, 0))), "DiffuseReflectance": fmt(m.get("diffuse", (0, 0, 0))),
                 "SpecularReflectance": fmt(m.get("specular", (0, 0, 0))), "MirrorReflectance": fmt(m.get("mirror", (0, 0, 0))),
                 "PhongExponent": repr(float(m.get("phong", 1.0))), "RefractionIndex": repr(float(m.get("refraction_index", 0.0))),
                 "AbsorptionIndex": repr(float(m.get("absorption_index", 0.0))),
                 "AbsorptionCoefficient": fmt(m.get("absorption_coef", (0, 0, 0)))}
            if m.get("roughness", 0.0) > 0:
                d["Roughness"] = repr(float(np.float32(m["roughness"])))
            if m.get("type", 0) in names:
                d["_type"] = names[m["type"]]
            mats.append(d)
        objects = {}
        mesh_nodes = []
        for k, m in enumerate(self.meshes):
            node = {"_id": str(k + 1), "Material": str(m["material"])}
            if m.get("smooth"):
                node["_shadingMode"] = "smooth"
            if m["verts"] is not None:
                ply = f"{self.name}_mesh{k + 1}.ply"
                write_ply(os.path.join(out_dir, ply), m["verts"], m["faces"])
                node["Faces"] = {"_plyFile": ply}
            else:
                node["Faces"] = {"_data": " ".join(str(int(v)) for v in np.asarray(m["faces"]).ravel())}
            if m.get("transform"):
                node["Transformations"] = tr_string(m["transform"])
            mesh_nodes.append(node)
        if mesh_nodes:
            objects["Mesh"] = mesh_nodes
        inst_nodes = []
        for k, inst in enumerate(self.instances):
            node = {"_id": str(len(self.meshes) + k + 1), "_baseMeshId": str(inst["base"] + 1),
                    "_resetTransform": "true" if inst.get("reset") else "false"}
            if inst.get("material"):
                node["Material"] = str(inst["material"])
            if inst.get("transform"):
                node["Transformations"] = tr_string(inst["transform"])
            inst_nodes.append(node)
        if inst_nodes:
            objects["MeshInstance"] = inst_nodes
        if self.spheres:
            objects["Sphere"] = []
            for k, s in enumerate(self.spheres):
                node = {"_id": str(k + 1), "Material": str(s["material"]), "Center": str(s["center"]), "Radius": repr(float(s["radius"]))}
                if s.get("transform"):
                    node["Transformations"] = tr_string(s["transform"])
                objects["Sphere"].append(node)
        if self.planes:
            objects["Plane"] = [{"_id": str(k + 1), "Material": str(p["material"]), "Point": str(p["point"]), "Normal": fmt(p["normal"])}
                                for k, p in enumerate(self.planes)]
        c = self.camera
        cam = {"_id": "1", "_type": "lookAt", "Position": fmt(c["position"]), "GazePoint": fmt(c["gaze_point"]), "Up": fmt(c["up"]),
               "FovY": repr(float(c["fovy"])), "NearDistance": repr(float(c["near"])),
               "ImageResolution": f"{c['width']} {c['height']}", "ImageName": c.get("name", self.name + ".png")}
        if c.get("num_samples", 1) > 1:
            cam["NumSamples"] = str(c["num_samples"])
        if c.get("aperture", 0.0) > 0:
            cam["ApertureSize"] = repr(float(c["aperture"]))
            cam["FocusDistance"] = repr(float(c["focus"]))
        scene = {"BackgroundColor": fmt(self.background), "ShadowRayEpsilon": repr(float(self.eps)),
                 "MaxRecursionDepth": str(self.max_depth), "Cameras": {"Camera": cam},
                 "Lights": {"AmbientLight": fmt(self.ambient),
                            "PointLight": [{"_id": str(k + 1), "Position": fmt(p), "Intensity": fmt(i)}
                                           for k, (p, i) in enumerate(self.point_lights)]},
                 "Materials": {"Material": mats},
                 "VertexData": {"_data": " ".join(fmt(v) for v in self.vertices), "_type": "xyz"},
                 "Objects": objects}
        if self.area_lights:
            scene["Lights"]["AreaLight"] = [{"_id": str(k + 1), "Position": fmt(p), "Normal": fmt(n), "Size": repr(float(np.float32(sz))),
                                             "Radiance": fmt(r)} for k, (p, n, sz, r) in enumerate(self.area_lights)]
        trans = {}
        if tr_tab["t"]:
            trans["Translation"] = [{"_id": str(k + 1), "_data": fmt(a)} for k, a in enumerate(tr_tab["t"])]
        if tr_tab["s"]:
            trans["Scaling"] = [{"_id": str(k + 1), "_data": fmt(a)} for k, a in enumerate(tr_tab["s"])]
        if tr_tab["r"]:
            trans["Rotation"] = [{"_id": str(k + 1), "_data": fmt(a)} for k, a in enumerate(tr_tab["r"])]
        if trans:
            scene["Transformations"] = trans
        path = os.path.join(out_dir, self.name + ".json")
        with open(path, "w") as f:
            json.dump({"Scene": scene}, f)
        return path


def write_ply(path, verts, faces0):
    """Binary little-endian PLY with the header shape parser.cpp:1126-1211 reads:
    float x,y,z vertices; faces as `uchar 3` + three int32 (0-based)."""
    verts = np.ascontiguousarray(verts, np.float32)
    faces0 = np.ascontiguousarray(faces0, np.int32)
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(verts)}\n"
                 "property float x\nproperty float y\nproperty float z\n"
                 f"element face {len(faces0)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(verts.tobytes())
        rec = np.zeros(len(faces0), dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"] = 3
        rec["i"] = faces0
        f.write(rec.tobytes())


# ------------------------------------------------------------------ workloads
def mesh_scene(grid=709, width=1920, height=1080, num_samples=1, name=None, tessellation="latlong"):
    """M2 (BASELINE.json configs[1]): one displaced-sphere mesh of 2*(grid-1)^2
    triangles (grid=709 -> 1 002 528) over a 2-triangle ground mesh, 2 point
    lights, diffuse materials, MaxRecursionDepth 0 => primary + 2 shadow rays per hit."""
    s = SynthScene(name or f"mesh_{tessellation}_{grid}" if tessellation != "latlong" else name or f"mesh_{grid}")
    s.max_depth = 0
    s.ambient = (25, 25, 25)
    s.vertices = [(-6, -1.15, -6), (6, -1.15, -6), (6, -1.15, 6), (-6, -1.15, 6)]
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.8, 0.7, 0.5), specular=(0.6, 0.6, 0.6), phong=30.0),
                   dict(ambient=(1, 1, 1), diffuse=(0.5, 0.6, 0.7), specular=(0.1, 0.1, 0.1), phong=5.0)]
    s.point_lights = [((4, 6, 5), (90000, 90000, 90000)), ((-5, 4, 2), (45000, 40000, 35000))]
    verts, faces = cube_sphere(grid) if tessellation == "cube" else displaced_sphere(grid)
    s.meshes = [dict(verts=None, faces=np.array([[1, 3, 2], [1, 4, 3]], np.int32), material=2),
                dict(verts=verts, faces=faces, material=1)]
    s.camera = dict(position=(0.4, 1.1, 4.2), gaze_point=(0, -0.05, 0), up=(0, 1, 0), fovy=32.0, near=1.0, width=width,
                    height=height, num_samples=num_samples)
    return s


def instanced_scene(grid=709, width=1920, height=1080, num_samples=1, max_depth=6, name=None, frame=0):
    """M3/M5: the displaced sphere plus 9 MeshInstances (translations, rotations,
    non-uniform and one negative-determinant scaling) with diffuse, mirror,
    conductor and dielectric materials, depth 6.
    frame > 0: a later frame of an animation of the same meshes: the instances turn and move,
    a light and the camera move, a material changes; from frame 2 on a sphere joins the top level; frame 3
    makes the mirror rough (the frame then needs the depth-first pixel loop)."""
    s = SynthScene(name or f"inst10_{grid}")
    s.max_depth = max_depth
    s.ambient = (20, 20, 20)
    s.background = (10, 12, 18)
    s.vertices = [(-14, -1.2, -14), (14, -1.2, -14), (14, -1.2, 14), (-14, -1.2, 14)]
    s.materials = [
        dict(ambient=(1, 1, 1), diffuse=(0.8, 0.7, 0.5), specular=(0.6, 0.6, 0.6), phong=30.0),
        dict(ambient=(1, 1, 1), diffuse=(0.45, 0.5, 0.55), specular=(0.1, 0.1, 0.1), phong=5.0),
        dict(type=ffi.RT_MAT_MIRROR, ambient=(0.2, 0.2, 0.2), diffuse=(0.1, 0.1, 0.1), specular=(0.5, 0.5, 0.5), mirror=(0.85, 0.85, 0.85), phong=60.0),
        dict(type=ffi.RT_MAT_DIELECTRIC, ambient=(0, 0, 0), diffuse=(0.02, 0.02, 0.02), specular=(0.3, 0.3, 0.3), mirror=(1, 1, 1),
             phong=80.0, refraction_index=1.5, absorption_coef=(0.02, 0.01, 0.03)),
        dict(type=ffi.RT_MAT_CONDUCTOR, ambient=(0.1, 0.1, 0.1), diffuse=(0.1, 0.08, 0.02), specular=(0.8, 0.7, 0.3), mirror=(1.0, 0.86, 0.57),
             phong=40.0, refraction_index=0.37, absorption_index=2.82),
    ]
    s.point_lights = [((6, 9, 8), (160000, 160000, 160000)), ((-8, 6, 3), (70000, 65000, 60000))]
    verts, faces = displaced_sphere(grid)
    s.meshes = [dict(verts=None, faces=np.array([[1, 3, 2], [1, 4, 3]], np.int32), material=2),
                dict(verts=verts, faces=faces, material=1)]
    rng = np.random.default_rng(20251205)
    mats = [3, 4, 5, 1, 3, 4, 5, 1, 4]
    for k in range(9):
        ang = float(rng.uniform(0, 360))
        axis = tuple(float(v) for v in rng.normal(size=3))
        sc = tuple(float(v) for v in rng.uniform(0.6, 1.3, size=3))
        if k == 4:
            sc = (-sc[0], sc[1], sc[2])           # negative determinant (raytracer.cpp:468)
        pos = (float(-5 + 2.5 * (k % 5) + rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.1, 0.5)), float(-3.0 - 2.8 * (k // 5) + rng.uniform(-0.3, 0.3)))
        s.instances.append(dict(base=1, material=mats[k], transform=[("s", sc), ("r", (ang,) + axis), ("t", pos)], reset=False))
    s.camera = dict(position=(0.5, 2.4, 7.5), gaze_point=(0, 0.0, -2.0), up=(0, 1, 0), fovy=40.0, near=1.0, width=width,
                    height=height, num_samples=num_samples)
    if frame:
        for k, inst in enumerate(s.instances):
            inst["transform"] = inst["transform"] + [("r", (9.0 * frame + k, 0.0, 1.0, 0.0)), ("t", (0.1 * frame, 0.05 * frame * (k % 3), 0.0))]
        s.point_lights[0] = ((6.0 - frame, 9.0, 8.0 + 0.5 * frame), (160000, 160000, 160000))
        s.materials[0]["diffuse"] = (0.8 - 0.1 * frame, 0.7, 0.5 + 0.1 * frame)
        s.camera["position"] = (0.5 + 0.4 * frame, 2.4, 7.5 - 0.2 * frame)
        if frame >= 2:
            s.spheres.append(dict(center=1, radius=0.6, material=3, transform=[("t", (14.5, 1.8, 12.0))]))
        if frame >= 3:
            s.materials[2]["roughness"] = 0.15
        s.name += f"_f{frame}"
    return s


def spheres_scene(width=256, height=256, num_samples=1, dof=False, name="spheres"):
    """hw1-style: spheres + a plane + triangles meshes, point lights, mirror (C1-like)."""
    s = SynthScene(name)
    s.max_depth = 4
    s.vertices = [(0, 0, 0), (-1.2, 0.1, -0.5), (1.3, 0.2, 0.3), (0, -1, 0), (-4, -1, -4), (4, -1, -4), (4, 3, -4), (-4, 3, -4)]
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.9, 0.3, 0.3), specular=(1, 1, 1), phong=50.0),
                   dict(type=ffi.RT_MAT_MIRROR, ambient=(0.1, 0.1, 0.1), diffuse=(0.2, 0.2, 0.2), specular=(1, 1, 1), mirror=(0.8, 0.8, 0.8), phong=100.0),
                   dict(ambient=(1, 1, 1), diffuse=(0.4, 0.6, 0.4), specular=(0.2, 0.2, 0.2), phong=3.0),
                   dict(type=ffi.RT_MAT_DIELECTRIC, diffuse=(0.01, 0.01, 0.01), specular=(0.5, 0.5, 0.5), mirror=(1, 1, 1), phong=100.0,
                        refraction_index=1.5, absorption_coef=(0.01, 0.05, 0.05))]
    s.point_lights = [((3, 5, 4), (50000, 50000, 50000)), ((-3, 2, 3), (20000, 20000, 30000))]
    s.spheres = [dict(center=1, radius=0.7, material=2), dict(center=2, radius=0.5, material=1),
                 dict(center=3, radius=0.6, material=4, transform=[("s", (1.0, 1.4, 0.8)), ("r", (30.0, 0, 0, 1))])]
    s.planes = [dict(point=4, normal=(0, 1, 0), material=3)]
    s.meshes = [dict(verts=None, faces=np.array([[5, 6, 7], [5, 7, 8]], np.int32), material=2)]
    s.camera = dict(position=(0.2, 0.8, 5.0), gaze_point=(0, 0, 0), up=(0, 1, 0), fovy=38.0, near=1.0, width=width, height=height,
                    num_samples=num_samples, aperture=0.15 if dof else 0.0, focus=5.0)
    return s


def glossy_area_scene(grid=48, width=320, height=200, num_samples=4, name=None):
    """shading that draws random numbers.  A displaced-sphere mesh (rough conductor), a rough
    dielectric instance of it, a rough mirror sphere and a diffuse floor under one area light and one point
    light, depth 4, a few samples per pixel.  Every pixel's generator is consumed in recursion order."""
    s = SynthScene(name or f"glossy_area_{grid}")
    s.max_depth = 4
    s.ambient = (20, 20, 20)
    s.background = (8, 10, 14)
    s.vertices = [(-8, -1.2, -8), (8, -1.2, -8), (8, -1.2, 8), (-8, -1.2, 8), (2.1, -0.4, 1.0)]
    s.materials = [
        dict(ambient=(1, 1, 1), diffuse=(0.6, 0.6, 0.55), specular=(0.2, 0.2, 0.2), phong=8.0),
        dict(type=ffi.RT_MAT_CONDUCTOR, ambient=(0.1, 0.1, 0.1), diffuse=(0.1, 0.08, 0.02), specular=(0.8, 0.7, 0.3),
             mirror=(1.0, 0.86, 0.57), phong=40.0, refraction_index=0.37, absorption_index=2.82, roughness=0.15),
        dict(type=ffi.RT_MAT_DIELECTRIC, ambient=(0, 0, 0), diffuse=(0.02, 0.02, 0.02), specular=(0.3, 0.3, 0.3), mirror=(1, 1, 1),
             phong=80.0, refraction_index=1.5, absorption_coef=(0.02, 0.01, 0.03), roughness=0.05),
        dict(type=ffi.RT_MAT_MIRROR, ambient=(0.2, 0.2, 0.2), diffuse=(0.1, 0.1, 0.1), specular=(0.5, 0.5, 0.5),
             mirror=(0.85, 0.85, 0.85), phong=60.0, roughness=0.3),
    ]
    s.point_lights = [((-5, 6, 4), (6000, 5600, 5200))]
    s.area_lights = [((1.0, 5.0, 2.0), (0.1, -1.0, -0.2), 1.5, (9000, 9000, 9600))]
    verts, faces = displaced_sphere(grid)
    s.meshes = [dict(verts=None, faces=np.array([[1, 3, 2], [1, 4, 3]], np.int32), material=1),
                dict(verts=verts, faces=faces, material=2, smooth=True)]
    s.instances = [dict(base=1, material=3, transform=[("s", (0.7, 0.9, 0.7)), ("r", (25.0, 0, 1, 0)), ("t", (-2.2, -0.1, 0.6))])]
    s.spheres = [dict(center=5, radius=0.8, material=4)]
    s.camera = dict(position=(0.3, 1.4, 6.5), gaze_point=(0, -0.1, 0), up=(0, 1, 0), fovy=34.0, near=1.0, width=width,
                    height=height, num_samples=num_samples)
    return s


def deep_chain_scene(n=64, width=160, height=120, name="deep_chain"):
    """A mesh whose midpoint-split BVH (raytracer.cpp:296-339) is a CHAIN about n levels deep: n self-similar
    triangles at x_k = 2^20 * 0.4^k facing the camera at the origin, each split peels off the largest one.  Every
    ray inside the cone walks the whole chain, left (= small, near) child first, with one pending sibling per level
    — 60-odd entries of the reference's stack[64], and ~3 per two levels of the device's one stack (rt_traverse.h),
    which is what the per-scene stack capacity exists for.  A rotated, scaled instance of the chain and a point
    light add transformed and any-hit walks of the same tree."""
    s = SynthScene(name)
    s.max_depth = 2
    s.background = (12, 24, 48)
    s.ambient = (30, 30, 30)
    x = (2.0 ** 20) * 0.4 ** np.arange(n, dtype=np.float64)
    v = np.stack([np.stack([x, -0.4 * x, -0.4 * x], -1), np.stack([x, 0.4 * x, -0.4 * x], -1), np.stack([x, 0 * x, 0.4 * x], -1)], 1)
    verts = v.reshape(-1, 3).astype(np.float32)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    s.vertices = [(0, 0, 0)]
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.8, 0.5, 0.3), specular=(0.5, 0.5, 0.5), phong=10.0),
                   dict(type=ffi.RT_MAT_MIRROR, ambient=(0.2, 0.2, 0.2), diffuse=(0.2, 0.3, 0.2), specular=(0.5, 0.5, 0.5), mirror=(0.6, 0.6, 0.6), phong=30.0)]
    s.point_lights = [((-3.0, 1.0, 2.0), (900, 800, 700))]
    s.meshes = [dict(verts=verts, faces=faces, material=1)]
    s.instances = [dict(base=0, material=2, transform=[("s", (0.5, 0.7, 0.7)), ("r", (35.0, 0, 0, 1)), ("t", (0.0, 0.0, 0.0))], reset=False)]
    s.camera = dict(position=(0.0, 0.0, 0.0), gaze_point=(1.0, 0.25, 0.0), up=(0, 0, 1), fovy=70.0, near=1.0, width=width, height=height, num_samples=1)
    return s
