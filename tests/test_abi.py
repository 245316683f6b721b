"""The C-ABI libraries load and export every symbol the headers declare (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", src)))


def test_rt_h_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "librt_hip.so"))
    names = declared("rt.h", "rt_")
    assert "rt_upload_scene" in names and "rt_render_tile" in names and len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n


def test_rt_host_h_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "librt_host.so"))
    names = declared("rt_host.h", "rth_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_no_device_fails_loudly():
    """Without a GPU the product must refuse, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from raytracer_amd import DeviceScene, RtError, scenes
    hs = scenes.spheres_scene(32, 32).build()
    with pytest.raises(RtError):
        DeviceScene(hs, 0)


def test_product_never_touches_oracle():
    """Nothing under raytracer_amd/ (nor include/) may import, link or name the oracle."""
    for base in ("raytracer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "librt_oracle" not in txt and "rto_render" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
