#!/usr/bin/env python3
"""Experiment: render the bench frame as S concurrent interleaved parts (S scene handles, S streams, S host
threads — the render calls are synchronous) — does overlapping the kernels of independent parts hide the
end-of-kernel tails?"""
import os, sys, time, threading
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, scenes
W, H = 1920, 1080
hs = (scenes.instanced_scene(709, W, H) if "inst" in sys.argv[1:] else scenes.mesh_scene(709, W, H)).build()
cam = hs.camera(0)
fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
ref = None
for S in (1, 2, 3, 4):
    dss = [DeviceScene(hs, 0) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def part(i):
        dss[i].render_rows_device(cam, 0, 0, W, H, S, i, fb.data_ptr(), W * 3, stream=streams[i].cuda_stream)
    def frame():
        th = [threading.Thread(target=part, args=(i,)) for i in range(1, S)]
        for t in th: t.start()
        part(0)
        for t in th: t.join()
    for _ in range(3): frame()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    img = fb.cpu()
    if ref is None: ref = img.clone()
    print(f"S={S}: {dt*1e3:.3f} ms/frame  identical={bool((img==ref).all())}", flush=True)
    for ds in dss: ds.close()
