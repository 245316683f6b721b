"""Frames per second of the frame driver (oracle/_ref/raytracer_hip = rt_main.cpp around the C ABI) on the three tap-water
frames of tests/golden/tap_frames and on a 12-frame synthetic animation over resident meshes: pipelined, --serial, and
one process per frame (what render_and_make_gif.sh does with the reference binary)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes
BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_hip")
TAP = os.path.join(ROOT, "tests", "golden", "tap_frames", "json")

def run(args, cwd):
    t = time.perf_counter()
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-2000:]
    return dt, [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]

def measure(tag, prefix, first, last, files):
    n = last - first + 1
    with tempfile.TemporaryDirectory() as d:
        run(["--frames", prefix, str(first), str(first)], d)            # warm the file cache and the GPU driver
        tp, lp = run(["--frames", prefix, str(first), str(last)], d)
        ts, ls = run(["--frames", prefix, str(first), str(last), "--serial"], d)
        t1 = 0.0
        for f in files:
            dt, _ = run([f], d)
            t1 += dt
    print(json.dumps({"animation": tag, "frames": n, "pipeline_wall_s": round(tp, 3), "pipeline_frames_per_s": round(n / tp, 3),
                      "serial_wall_s": round(ts, 3), "serial_frames_per_s": round(n / ts, 3),
                      "one_process_per_frame_wall_s": round(t1, 3), "one_process_per_frame_frames_per_s": round(n / t1, 3),
                      "driver_summary_pipeline": lp[-1], "driver_summary_serial": ls[-1]}), flush=True)

measure("tap_water 0000..0002 (1000x1000, 64 spp, new water mesh per frame: upload each frame)", os.path.join(TAP, "tap_0"), 0, 2,
        [os.path.join(TAP, f"tap_{k:04d}.json") for k in range(3)])
with tempfile.TemporaryDirectory() as src:
    files = []
    for f in range(12):
        spec = scenes.instanced_scene(grid=256, width=1280, height=720, frame=f)
        js = spec.write_json(src)
        dst = os.path.join(src, f"anim_{f:03d}.json")
        os.rename(js, dst)
        files.append(dst)
    measure("synthetic inst10 (10 x 130 050 triangles, 1280x720, depth 6, resident meshes: rt_update_scene)", os.path.join(src, "anim_"), 0, 11, files)
