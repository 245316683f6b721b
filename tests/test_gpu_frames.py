"""SURVEY §8f-4 — frames of an animation through the drop-in driver (raytracer_amd/host/rt_main.cpp built against
the reference into oracle/_ref/raytracer_hip: the reference's parser, BVH builders and PNG writer around the C ABI).
The driver keeps the device scene resident across frames (rt_update_scene when the meshes are the uploaded ones) and
overlaps parse+build of frame k+1, render of frame k and PNG encoding of frame k-1 on host threads; --gpus N sends
each GPU's row bands to GPU 0 with one strided peer copy.  Inputs are files only (JSON + PLY): the author's first
three tap-water frames (tests/golden/tap_frames, with crops of the author's own renders) and synthetic frames
written in the reference's input format."""
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_hip")
TAP = os.path.join(util.GOLDEN, "tap_frames")


def _run(args, cwd, env=None, timeout=600):
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/raytracer_hip not built (needs /root/reference at build time)")
def test_tap_water_frames_equal_the_authors_renders(tmp_path):
    """tap_0000..tap_0002 (1000x1000, 64 spp, rough conductor + dielectric: the depth-first pixel loop; the water mesh
    differs per frame, so every frame is a fresh upload) through the pipeline; the PNGs the driver writes must equal
    the author's committed renders on the fixture's crop, byte for byte."""
    lines = _run(["--frames", os.path.join(TAP, "json", "tap_0"), "0", "2"], str(tmp_path))
    frames = [l for l in lines if "camera" in l]
    assert [l["camera"] for l in frames] == ["tap_0000.png", "tap_0001.png", "tap_0002.png"]
    assert lines[-1]["frames"] == 3 and lines[-1]["pipeline"] is True
    for k in range(3):
        got = np.array(Image.open(tmp_path / f"tap_{k:04d}.png").convert("RGB"))
        want = np.array(Image.open(os.path.join(TAP, f"tap_{k:04d}_author_crop_300_400_400x200.png")).convert("RGB"))
        assert got.shape == (1000, 1000, 3)
        assert (got[400:600, 300:700] == want).all(), k
    print("tap frames:", json.dumps(lines[-1]))


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/raytracer_hip not built (needs /root/reference at build time)")
def test_resident_scene_pipeline_multi_gpu_bands_and_tiles(tmp_path):
    """Four frames of an animation over the same meshes (transforms, a light, a material and the camera change; a
    sphere joins; the last frame needs the depth-first pixel loop), written as JSON + PLY: the first frame is an
    upload, the others rt_update_scene; every PNG equals the oracle's render of that frame.  The same frames with
    --serial, with --gpus 2 on one device (the band plan and the strided device-to-device copies of the multi-GPU
    path) and with --tile give the same files."""
    from oracle import binding as ob
    from raytracer_amd import scenes
    src = tmp_path / "in"
    src.mkdir()
    want = []
    for f in range(4):
        spec = scenes.instanced_scene(grid=48, width=328, height=184, frame=f)
        js = spec.write_json(str(src))
        os.rename(js, src / f"anim_{f:03d}.json")
        hs = spec.build()
        want.append(ob.render(hs, hs.camera(0))[0])
    runs = {"pipeline": [], "serial": ["--serial"], "gpus2": ["--gpus", "2"], "tiles": ["--tile", "100x64"], "gpus3_tiles": ["--gpus", "3", "--tile", "96x40"]}
    for name, extra in runs.items():
        out = tmp_path / name
        out.mkdir()
        lines = _run(["--frames", str(src / "anim_"), "0", "3"] + extra, str(out), env={"RT_MAIN_ONE_DEVICE": "1"})
        frames = [l for l in lines if "camera" in l]
        assert [l["scene_on_device"] for l in frames] == ["upload", "update", "update", "update"], name
        assert lines[-1]["pipeline"] is (name != "serial")
        pngs = sorted(p for p in os.listdir(out) if p.endswith(".png"))
        assert len(pngs) == 4
        for f, p in enumerate(pngs):
            got = np.array(Image.open(out / p).convert("RGB"))
            assert (got == want[f]).all(), (name, f, int((got != want[f]).any(-1).sum()))
        print(name, json.dumps(lines[-1]))


def test_bench_multi_gpu_branch_runs_over_rccl_with_one_rank():
    """bench.py's N > 1 branch (strong-scaling frame, barrier, all-reduces of the clocks and ray counts, gather of the
    RGB8 rows to rank 0) launched as the driver launches it — python -m torch.distributed.run, backend nccl = RCCL —
    with ONE rank (RT_BENCH_FORCE_DIST=1: a one-rank communicator, the gather not short-circuited), so that RCCL
    initialisation and a dist.gather of uint8 rows execute on real hardware on a 1-GPU box.  The gathered frame's
    checksum in the line must be that of the same frame rendered directly."""
    import hashlib
    import sys
    import raytracer_amd as rt
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    env.pop("RT_BENCH_REHEARSE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "96", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["gather"] == "rccl"
    assert line["value"] > 0 and line["steps"] == 2
    hs = rt.scenes.instanced_scene(96, 1920, 1080).build()      # (--grid 96: the strong-scaling frame at its rehearsal size)
    ds = rt.DeviceScene(hs, 0)
    rgb, _ = ds.render(hs.camera(0))
    ds.close()
    assert line["config"]["frame_sha16"] == hashlib.sha256(np.ascontiguousarray(rgb).tobytes()).hexdigest()[:16]
    assert abs(line["config"]["rays_per_frame"] - 0) > 0
