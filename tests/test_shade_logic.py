"""The device's ray generation and shading arithmetic (raytracer_amd/csrc/rt_shade.h: ComputeRay with and without
depth of field, both Fresnel terms, ComputeDiffuseAndSpecular with its powf, CreateOrthonormalBasis, PerturbReflection —
the text the HIP kernels compile) built for the host and compared bit for bit with the oracle's restatement of the
reference (raytracer.cpp:343-378, 1438-1537) on random and edge operands."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shading_arithmetic_equals_oracle_bit_for_bit(tmp_path):
    clang = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")
    if clang is None:
        pytest.skip("clang++ not found (ext_vector_type needs clang)")
    exe = str(tmp_path / "shade_logic")
    subprocess.run([clang, "-O1", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "shim"), "-Wno-everything",
                    "-o", exe, os.path.join(ROOT, "tests", "shade_logic.cpp"), "-lm", "-fopenmp"], check=True, cwd=ROOT)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "total: 0 mismatches" in out.stdout, out.stdout[-2000:]
