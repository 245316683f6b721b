"""raytracer_amd/csrc/rt_libm.h — the device's powf / expf, a restatement of glibc 2.35's algorithm (the library the
reference calls through std::pow / std::exp, raytracer.cpp:1356-1358,1428,1517) — compiled for the host and compared
bit for bit with the C library of this box over 2.4e8 operands: the Blinn-Phong and Beer's-law ranges densely, every
37th float bit pattern for expf, random bit patterns of both powf arguments, and the special values."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_powf_expf_equal_glibc_bit_for_bit(tmp_path):
    exe = str(tmp_path / "libm_logic")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fopenmp", "-DRT_LIBM_HOST", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "libm_logic.cpp"), "-lm"], check=True, cwd=ROOT)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "total: 0 mismatches" in out.stdout, out.stdout[-3000:]
    n = int([l for l in out.stdout.splitlines() if l.startswith("operands:")][0].split()[1])
    assert n >= 100_000_000
