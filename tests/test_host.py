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


def test_damaged_blobs_are_refused_not_trusted(tmp_path):
    """the RTSCENE1 loader must not believe counts the file cannot back (a truncated blob, a count of
    2^32 elements) nor accept top-level arrays of different lengths; it fails with an error, never aborts."""
    import struct
    from raytracer_amd import HostScene, RtError, scenes
    good = tmp_path / "good.rtscene"
    scenes.spheres_scene(32, 32).build().save(str(good))
    blob = good.read_bytes()
    HostScene.load(str(good))
    cut = tmp_path / "cut.rtscene"
    cut.write_bytes(blob[: len(blob) // 2])
    with pytest.raises(RtError):
        HostScene.load(str(cut))
    # header: magic 8 + 12 sizes 48 + colours 24 + eps/depth 8 = 88; then the count of top_nodes
    huge = tmp_path / "huge.rtscene"
    huge.write_bytes(blob[:88] + struct.pack("<Q", 1 << 32) + blob[96:])
    with pytest.raises(RtError):
        HostScene.load(str(huge))
    n_top_nodes = struct.unpack("<Q", blob[88:96])[0]
    off = 96 + 32 * n_top_nodes                      # count of top_prim_idx
    n_idx = struct.unpack("<Q", blob[off:off + 8])[0]
    assert n_idx >= 2
    short = tmp_path / "short.rtscene"                # one index fewer than top_prims
    short.write_bytes(blob[:off] + struct.pack("<Q", n_idx - 1) + blob[off + 8:off + 8 + 4 * (n_idx - 1)] + blob[off + 8 + 4 * n_idx:])
    with pytest.raises(RtError):
        HostScene.load(str(short))


def test_rows_partition_is_exact():
    from raytracer_amd import multigpu
    for world in (1, 2, 3, 8):
        for h in (8, 9, 1080, 1528, 3055, 4320):
            got = np.concatenate([multigpu.rows_of(r, world, h).numpy() for r in range(world)])
            assert sorted(got.tolist()) == list(range(h))


def test_row_plan_of_the_8k_frame_over_8_ranks():
    """BASELINE configs[4]: 4320 rows = 540 eight-row bands over 8 ranks.  The plan bench.py --gpus 8 and
    rt_main.cpp --gpus 8 follow: rank r owns the bands r, r + 8, ... (rt_render_tile_rows_device's row_part), 67 or
    68 bands each, every row exactly once, and the gather pads every rank's share to the largest."""
    from raytracer_amd import multigpu
    H, world = 4320, 8
    shares = [multigpu.rows_of(r, world, H).numpy() for r in range(world)]
    bands = [np.unique(s // 8) for s in shares]
    for r in range(world):
        assert (bands[r] % world == r).all()
        assert len(shares[r]) == 8 * len(bands[r])                      # 4320 is a multiple of 8: whole bands only
        assert len(bands[r]) in (67, 68)
    assert sum(len(s) for s in shares) == H and len(np.unique(np.concatenate(shares))) == H
    assert multigpu.max_rows(world, H) == 68 * 8
    # bytes on the wire per frame: each rank sends its padded share of RGB8 rows once
    assert multigpu.max_rows(world, H) * 7680 * 3 == 12_533_760


def _bad_bvh_scene(kind):
    """A valid small scene whose mesh-BVH node array is then altered in place (a copy the description points at)."""
    import ctypes as C
    from raytracer_amd import ffi, scenes
    hs = scenes.mesh_scene(grid=12, width=32, height=32).build()
    src = ffi.SceneDesc.from_address(hs.desc_ptr)
    desc = ffi.SceneDesc()
    C.memmove(C.byref(desc), hs.desc_ptr, C.sizeof(desc))
    nodes = (ffi.BvhNode * src.n_mesh_nodes)()
    C.memmove(nodes, src.mesh_nodes, C.sizeof(nodes))
    mb = src.mesh_bvhs[1]                      # the sphere mesh (BVH 0 is the two-triangle ground)
    base, n = mb.node_offset, mb.node_count
    interior = [i for i in range(base, base + n) if nodes[i].prim_count == 0]
    assert len(interior) >= 4
    if kind == "shared_child":                 # two interior nodes name the same pair of children (a DAG)
        nodes[interior[2]].left_or_first = nodes[interior[3]].left_or_first
    elif kind == "orphan":                     # an interior node becomes a leaf: its children lose their parent
        nodes[interior[1]].prim_count = 1
        nodes[interior[1]].left_or_first = 0
    elif kind == "backward_child":
        nodes[interior[3]].left_or_first = 0
    desc.mesh_nodes = C.cast(nodes, C.POINTER(ffi.BvhNode))
    return hs, desc, nodes


@pytest.mark.parametrize("kind", ["shared_child", "orphan", "backward_child"])
def test_malformed_bvh_is_rejected_at_upload(kind):
    """a BVH that is not a tree over all its nodes (shared children, unreferenced nodes) would make the
    device-side collapse write out of bounds; rt_upload_scene must refuse it with RT_ERR_INVALID.  Validation runs
    before the device is looked for, so this holds on a box without a GPU as well."""
    import ctypes as C
    from raytracer_amd import ffi
    hs, desc, keep = _bad_bvh_scene(kind)
    L = ffi.hip_lib()
    h = C.c_void_p()
    rc = L.rt_upload_scene(0, C.byref(desc), C.byref(h))
    assert rc == -1, (rc, L.rt_last_error().decode())           # RT_ERR_INVALID
    assert h.value is None
    msg = L.rt_last_error().decode()
    assert "mesh BVH" in msg, msg


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from raytracer_amd import multigpu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W = 24
for H in (52, 48, 8, 16, 4320 // 8):          # ragged last band (index path); whole bands (strided views); fewer bands than ranks
    want = (torch.arange(H * W * 3, dtype=torch.int64).reshape(H, W, 3) % 251).to(torch.uint8)
    for rep in range(2):                      # the plan and its buffers are reused
        fb = torch.full((H, W, 3), 255, dtype=torch.uint8)           # garbage in rows this rank does not own
        mine = multigpu.rows_of(rank, world, H)
        fb[mine] = want[mine]
        out = multigpu.gather_frame(fb, rank, world)
        if rank == 0:
            assert out is not None and torch.equal(out, want), f"merged frame differs (H={H})"
        else:
            assert out is None
if rank == 0:
    print("MERGE_OK")
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_gather_frame_gloo(tmp_path, world):
    """The N>1 path's only collective (framebuffer row gather), 2 and 3 processes over gloo on CPU."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = str(29517 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MERGE_OK" in r.stdout
