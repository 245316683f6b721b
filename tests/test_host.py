"""Host-side logic: blob round trip, builder invariants, multi-GPU partition (gloo, world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_blob_roundtrip(tmp_path):
    from raytracer_amd import HostScene, scenes
    hs = scenes.instanced_scene(grid=16, width=64, height=48).build()
    p1, p2 = tmp_path / "a.rtscene", tmp_path / "b.rtscene"
    hs.save(str(p1))
    HostScene.load(str(p1)).save(str(p2))
    assert p1.read_bytes() == p2.read_bytes()
    assert hs.camera(0).image_width == 64 and hs.num_cameras == 1


def test_load_missing_blob_raises(tmp_path):
    from raytracer_amd import HostScene, RtError
    with pytest.raises(RtError):
        HostScene.load(str(tmp_path / "nope.rtscene"))
    bad = tmp_path / "bad.rtscene"
    bad.write_bytes(b"not a scene")
    with pytest.raises(RtError):
        HostScene.load(str(bad))


def test_rows_partition_is_exact():
    from raytracer_amd import multigpu
    for world in (1, 2, 3, 8):
        for h in (8, 9, 1080, 1528, 3055):
            got = np.concatenate([multigpu.rows_of(r, world, h).numpy() for r in range(world)])
            assert sorted(got.tolist()) == list(range(h))


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from raytracer_amd import multigpu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
H, W = 52, 24
want = (torch.arange(H * W * 3, dtype=torch.int64).reshape(H, W, 3) % 251).to(torch.uint8)
fb = torch.full((H, W, 3), 255, dtype=torch.uint8)           # garbage in rows this rank does not own
mine = multigpu.rows_of(rank, world, H)
fb[mine] = want[mine]
out = multigpu.gather_frame(fb, rank, world)
if rank == 0:
    assert out is not None and torch.equal(out, want), "merged frame differs"
    print("MERGE_OK")
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
'''


def test_gather_frame_gloo_world2(tmp_path):
    """The N>1 path's only collective (framebuffer row gather), 2 processes over gloo on CPU."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script), ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MERGE_OK" in r.stdout
