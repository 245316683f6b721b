"""The SAH rebuild behind RT_UPLOAD_REBUILD_MESH_BVH (raytracer_amd/csrc/rt_rebuild.h, host code of librt_hip.so)
built on its own and checked structurally: the trees must be valid input for the parity path's upload (reference
node format, raytracer.cpp:296-339 / Bvh.h:39-49), hold every face exactly once, and carry each triangle's
position in the caller's leaf order (the tie rule of the near-first kernels)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rebuilt_trees_are_valid_reference_format(tmp_path):
    exe = str(tmp_path / "rebuild_logic")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "rebuild_logic.cpp")], check=True, cwd=ROOT)
    for threads in ("8", "1"):      # subtrees grown side by side and spliced / the serial build
        out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, RT_REBUILD_THREADS=threads))
        assert out.returncode == 0, out.stdout[-3000:]
        assert "total: 0 failures" in out.stdout, out.stdout[-3000:]
