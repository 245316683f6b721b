"""The per-pixel random number generator of the kernels (raytracer_amd/csrc/rt_mt.h: mt19937 made word by word, its
first block straight from the seeding recurrence, the canonical float, std::shuffle's pair path) built for the host and
compared with libstdc++'s std::mt19937 / std::shuffle / std::uniform_real_distribution<float> — what the reference's
pixel loop calls (raytracer.cpp:44-69) — over 160 seeds, 3200 draws each (five 624-word blocks) and 14 sample counts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_shuffle_and_floats_equal_libstdcxx(tmp_path):
    cxx = shutil.which("g++") or shutil.which("clang++", path="/opt/rocm/lib/llvm/bin")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "mt_logic")
    subprocess.run([cxx, "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "tests", "shim"), "-w",
                    "-o", exe, os.path.join(ROOT, "tests", "mt_logic.cpp")], check=True, cwd=ROOT)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "total: 0 mismatches" in out.stdout, out.stdout[-2000:]
