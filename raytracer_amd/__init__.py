"""raytracer_amd — MI355X-native per-pixel ray/scene intersection path.

The product is raytracer_amd/librt_hip.so (C ABI: include/rt.h).  This package
is thin ctypes plumbing around it plus the host helper library.
"""
from .ffi import (CameraDesc, DeviceScene, HostScene, Material, RtError, SceneBuilder, Stats,  # noqa: F401
                  RT_FLAG_NEAR_FIRST, RT_FLAG_ONE_LAUNCH, RT_FLAG_PER_BOUNCE_LAUNCHES, RT_FLAG_SEPARATE_LAUNCHES, RT_UPLOAD_REBUILD_MESH_BVH, RT_UPLOAD_REBUILD_ON_DEVICE, RT_UPLOAD_REBUILD_LBVH, RT_FLAG_TIMING, RT_OPT_CHUNK_SAMPLES, RT_OPT_FLAGS, RT_OPT_POOL_RECORDS, RT_OPT_STACK_CAP,
                  RT_KIND_MESH, RT_KIND_PLANE, RT_KIND_SPHERE, RT_KIND_TRIANGLE,
                  RT_MAT_CONDUCTOR, RT_MAT_DIELECTRIC, RT_MAT_MIRROR, RT_MAT_NONE,
                  hip_lib, host_lib, mat_mul, mat_rotation, mat_scaling, mat_translation)
from . import multigpu, scenes  # noqa: E402,F401
