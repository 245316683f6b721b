"""The device's ray/box and ray/primitive arithmetic (raytracer_amd/csrc/rt_box.h: box_test, node_slabs4, tri_test,
sphere_test, plane_test — the text the HIP kernels compile) built for the host and compared with the oracle's restatement of the reference's IntersectAABB
(raytracer.cpp:815-894) over adversarial operands: zero, negative-zero and denormal direction components (1/d
infinite of either sign), origins exactly on box planes, degenerate boxes, infinite and negative bounds.  These
cases cannot be reached through camera rays in a GPU test, and they decide whether a leaf is visited."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_box_arithmetic_equals_reference_on_adversarial_operands(tmp_path):
    clang = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")
    if clang is None:
        pytest.skip("clang++ not found (ext_vector_type needs clang)")
    exe = str(tmp_path / "box_logic")
    subprocess.run([clang, "-O1", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "shim"), "-Wno-everything",
                    "-o", exe, os.path.join(ROOT, "tests", "box_logic.cpp"), "-lm", "-fopenmp"], check=True, cwd=ROOT)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "total: 0 mismatches" in out.stdout, out.stdout[-2000:]
