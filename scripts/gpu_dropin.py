#!/usr/bin/env python3
"""Exercise oracle/_ref/raytracer_hip (the reference's main() with the loop replaced by the C ABI) on the GPU box:
write a synthetic scene as JSON+PLY, run --backend hip and --backend cpu, compare the two PNGs byte for byte."""
import os, subprocess, sys, tempfile, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from raytracer_amd import scenes
exe = os.path.join(ROOT, "oracle", "_ref", "raytracer_hip")
with tempfile.TemporaryDirectory() as td:
    for spec in (scenes.instanced_scene(grid=128, width=960, height=540), scenes.mesh_scene(709, 1920, 1080)):
        js = spec.write_json(td)
        outs = {}
        for backend in ("hip", "cpu", "hip-fast"):
            args = ["--backend", "hip", "--near-first", "--rebuild-bvh"] if backend == "hip-fast" else ["--backend", backend]
            r = subprocess.run([exe, js] + args, cwd=td, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            info = json.loads(line)
            outs[backend] = (np.array(Image.open(os.path.join(td, info["camera"])).convert("RGB")), info)
            os.remove(os.path.join(td, info["camera"]))
        same = bool((outs["hip"][0] == outs["cpu"][0]).all())
        print(spec.name, "png identical:", same, "| hip", outs["hip"][1]["render_s"], "s", outs["hip"][1]["mrays_per_s"], "Mrays/s (incl. PCIe copy-back) | cpu", outs["cpu"][1]["render_s"], "s", flush=True)
        assert same
        fast = outs["hip-fast"]
        d = np.abs(fast[0].astype(np.int16) - outs["cpu"][0].astype(np.int16)).max(axis=-1)
        print("   --near-first --rebuild-bvh:", fast[1]["render_s"], "s", fast[1]["mrays_per_s"], "Mrays/s, upload", fast[1]["upload_s"], "s; pixels equal to the CPU render",
              float((d == 0).mean()), "within 1 LSB", float((d <= 1).mean()), flush=True)
        assert (d <= 1).mean() >= 0.999
