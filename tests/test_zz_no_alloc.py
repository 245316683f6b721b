"""rt_render_tile_device_noalloc (include/rt.h): the render entry point that allocates nothing — all its work
memory is a static array inside the library, one copy per GPU.  Called through the C ABI and compared, byte for byte
and id for id, with rt_render_tile_device on the same tile: two calls in a row, the second on a smaller input (the
array keeps the first call's contents), on a scene with child rays (chunk loop, pool-overflow retries in the fixed
pool), a multi-sample camera (generator state in the array) and a depth-0 mesh (asynchronous return); then the
limits.  This file is collected last: it must pass alone in a fresh process and after all other tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(rt, ds, cam, x0, y0, w, h):
    import torch
    dev = "cuda:0"
    a = torch.full((h, w, 3), 3, dtype=torch.uint8, device=dev)
    ha = torch.full((h, w, 2), 3, dtype=torch.int32, device=dev)
    b = torch.full((h, w, 3), 5, dtype=torch.uint8, device=dev)
    hb = torch.full((h, w, 2), 5, dtype=torch.int32, device=dev)
    ds.render_device_noalloc(cam, x0, y0, w, h, a.data_ptr(), w * 3, d_hits_ptr=ha.data_ptr())
    ds.sync()
    ds.render_device(cam, x0, y0, w, h, b.data_ptr(), w * 3, d_hits_ptr=hb.data_ptr(), stats=rt.Stats())
    assert torch.equal(a, b), int((a != b).any(-1).sum().item())
    assert torch.equal(ha, hb)
    return a.cpu().numpy()


def test_no_alloc_entry_point_equals_render_tile_device():
    import raytracer_amd as rt
    from oracle import binding as ob
    rt.hip_lib()
    # depth 6, two lights: children and shadow slots in the fixed pool; 600x400 does not fit one chunk of it
    hs = rt.scenes.instanced_scene(grid=64, width=640, height=400).build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    big = _both(rt, ds, cam, 16, 0, 600, 400)
    small = _both(rt, ds, cam, 200, 120, 77, 45)                     # second call, smaller input, same array
    assert (small == big[120:165, 184:261]).all()
    o_rgb, _, _ = ob.render(hs, cam, x0=200, y0=120, w=77, h=45)
    assert (small == o_rgb).all()
    ds.close()
    # multi-sample with depth of field: MT19937 state and the stratification permutation live in the array
    hs = rt.scenes.spheres_scene(200, 150, num_samples=16, dof=True).build()
    ds = rt.DeviceScene(hs, 0)
    _both(rt, ds, hs.camera(0), 0, 0, 200, 150)
    _both(rt, ds, hs.camera(0), 31, 17, 50, 40)
    ds.close()
    # depth 0 (returns before the kernels ran), another scene on the same GPU's array
    hs = rt.scenes.mesh_scene(grid=128, width=512, height=288).build()
    ds = rt.DeviceScene(hs, 0)
    _both(rt, ds, hs.camera(0), 0, 0, 512, 288)
    _both(rt, ds, hs.camera(0), 100, 100, 64, 8)
    # limits: too many samples for the entry point; a scene that draws random numbers while shading
    import torch
    hs_big = rt.scenes.mesh_scene(grid=16, width=4096, height=2048).build()
    dsb = rt.DeviceScene(hs_big, 0)
    fb = torch.zeros((2048, 4096, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(rt.RtError, match="exceed the maximum"):
        dsb.render_device_noalloc(hs_big.camera(0), 0, 0, 4096, 2048, fb.data_ptr(), 4096 * 3)
    dsb.close()
    hs_g = rt.scenes.glossy_area_scene(grid=16, width=64, height=40, num_samples=1).build()
    dsg = rt.DeviceScene(hs_g, 0)
    with pytest.raises(rt.RtError, match="random numbers"):
        dsg.render_device_noalloc(hs_g.camera(0), 0, 0, 64, 40, fb.data_ptr(), 64 * 3)
    dsg.close()
    _both(rt, ds, hs.camera(0), 0, 0, 512, 288)                      # refused calls left nothing behind
    ds.close()


def test_no_alloc_entry_point_allocates_nothing_and_can_be_captured_in_a_hip_graph():
    """What defines the entry point: device memory in use does not move across its calls (child rays, multi-sample and
    depth-0 scenes), and a depth-0 call can be recorded with stream capture in the mode that turns every allocation
    into an error, replayed, and gives rt_render_tile_device's bytes."""
    import torch
    import raytracer_amd as rt
    rt.hip_lib()
    dev = "cuda:0"
    cases = [(rt.scenes.instanced_scene(grid=64, width=640, height=400), (16, 0, 600, 400)),
             (rt.scenes.spheres_scene(200, 150, num_samples=16, dof=True), (0, 0, 200, 150)),
             (rt.scenes.mesh_scene(grid=128, width=512, height=288), (0, 0, 512, 288))]
    for spec, (x0, y0, w, h) in cases:
        hs = spec.build()
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        hits = torch.zeros((h, w, 2), dtype=torch.int32, device=dev)
        ds.render_device_noalloc(cam, x0, y0, w, h, out.data_ptr(), w * 3, d_hits_ptr=hits.data_ptr())   # warm-up (code object, first touch)
        ds.sync()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(3):
            ds.render_device_noalloc(cam, x0, y0, w, h, out.data_ptr(), w * 3, d_hits_ptr=hits.data_ptr())
            ds.render_device_noalloc(cam, x0 + 8, y0 + 8, w // 2, h // 2, out.data_ptr(), w * 3)
        ds.sync()
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, f"device memory in use moved by {free0 - free1} bytes across no-allocation calls"
        ds.close()
    # stream capture, depth-0 scene (the last one built above): global capture mode = any hipMalloc / synchronisation fails
    hs = cases[2][0].build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    w, h = 512, 288
    ref = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    ds.render_device(cam, 0, 0, w, h, ref.data_ptr(), w * 3, stats=rt.Stats())
    out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    ds.render_device_noalloc(cam, 0, 0, w, h, out.data_ptr(), w * 3)     # (nothing left to set up inside the capture)
    ds.sync()
    out.zero_()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="global"):
        ds.render_device_noalloc(cam, 0, 0, w, h, out.data_ptr(), w * 3, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(out.sum().item()) == 0, "recording a graph must not run anything"
    for _ in range(2):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ds.sync()                                                          # reports what the replay's kernels left
        assert torch.equal(out, ref)
    ds.close()
