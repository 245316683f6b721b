"""Probe: the one-launch pipeline (default) against the per-bounce pipeline (RT_FLAG_PER_BOUNCE_LAUNCHES) and the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
from oracle import binding as ob

def run(tag, spec, oracle=True, reps=3):
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    out = {}
    for name, flags in (("mega", rt.RT_FLAG_ONE_LAUNCH), ("bounce", rt.RT_FLAG_PER_BOUNCE_LAUNCHES)):
        ds.set_option(rt.RT_OPT_FLAGS, flags | rt.RT_FLAG_TIMING)
        st = rt.Stats()
        rgb, hits = ds.render(cam, want_hits=True, stats=st)
        t = []
        for _ in range(reps):
            st = rt.Stats()
            ds.render(cam, stats=st)
            t.append(st.ms_total)
        out[name] = (rgb, hits, st, min(t))
        rays = st.rays_primary + st.rays_secondary + st.rays_shadow
        print(f"{tag:28s} {name:7s} {min(t):8.3f} ms {rays / min(t) / 1e3:8.1f} Mrays/s  rays {st.rays_primary} {st.rays_secondary} {st.rays_shadow} launches {st.launches_closest}+{st.launches_shadow} chunks {st.chunks} | closest {st.ms_closest:.3f} shade {st.ms_shade:.3f} shadow {st.ms_shadow:.3f} combine {st.ms_combine:.3f}", flush=True)
    a, b = out["mega"], out["bounce"]
    print(f"{tag:28s} mega == bounce: pixels {(a[0] == b[0]).all()} hits {(a[1] == b[1]).all()} rays {[a[2].rays_primary, a[2].rays_secondary, a[2].rays_shadow] == [b[2].rays_primary, b[2].rays_secondary, b[2].rays_shadow]}", flush=True)
    if oracle:
        o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
        print(f"{tag:28s} mega == oracle: pixels differ {int((a[0] != o_rgb).any(-1).sum())} hits differ {int((a[1] != o_hits).any(-1).sum())}", flush=True)
    ds.close()

which = sys.argv[1:] or ["small"]
if "small" in which:
    run("spheres 200x150", scenes.spheres_scene(200, 150))
    run("mesh g96 640x360", scenes.mesh_scene(grid=96, width=640, height=360))
    run("inst g64 480x270", scenes.instanced_scene(grid=64, width=480, height=270))
    run("spheres 16spp dof", scenes.spheres_scene(200, 150, num_samples=16, dof=True))
    run("deep chain", scenes.deep_chain_scene())
if "mid" in which:
    for w, h in ((480, 270), (960, 540), (1280, 720), (1920, 1080)):
        run(f"mesh1m {w}x{h}", scenes.mesh_scene(709, w, h), oracle=False, reps=8)
        run(f"inst10 {w}x{h}", scenes.instanced_scene(709, w, h), oracle=False, reps=5)
        run(f"spheres {w}x{h}", scenes.spheres_scene(w, h), oracle=False, reps=8)
    run("mesh g96 4spp 640x360", scenes.mesh_scene(grid=96, width=640, height=360, num_samples=4), oracle=False, reps=8)
if "hd" in which:
    run("mesh g96 640x360", scenes.mesh_scene(grid=96, width=640, height=360))
    run("mesh1m 1280x720", scenes.mesh_scene(709, 1280, 720), oracle=False, reps=8)
    run("mesh1m 1920x1080", scenes.mesh_scene(709, 1920, 1080), oracle=False, reps=8)
    run("inst10 1920x1080", scenes.instanced_scene(709, 1920, 1080), oracle=False, reps=5)
    run("mesh1m 4K", scenes.mesh_scene(709, 3840, 2160), oracle=False, reps=4)
if "big" in which:
    run("mesh1m 1080p", scenes.mesh_scene(709, 1920, 1080), oracle=False, reps=10)
    run("inst10 1080p", scenes.instanced_scene(709, 1920, 1080), oracle=False, reps=5)
    run("mesh1m 4K", scenes.mesh_scene(709, 3840, 2160), oracle=False, reps=5)
    run("inst10 8K", scenes.instanced_scene(709, 7680, 4320), oracle=False, reps=3)
