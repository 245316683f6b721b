"""GPU parity tests proper: the HIP path, called through the C ABI, against
(a) the reference's own renders and hit identities (tests/golden), (b) the
oracle on seeded synthetic scenes, (c) size-independent properties at the
bench's full size.  Bars: images byte-exact (north_star allows 1 LSB on 99.9 %);
primary hit ids identical (north_star allows 99.99 %)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

GOLD = ["hw1_simple", "hw1_spheres_mirror", "hw1_cornellbox_recursive", "hw1_two_spheres", "hw1_spheres_with_plane",
        "hw2_mirror_room", "hw2_metal_glass_plates", "hw2_two_berserkers", "hw2_ellipsoids", "hw2_simple_transform",
        # the scanned 100 000-face mesh (deep midpoint tree) and a tile of the scene whose top-level
        # BVH is a single 13-prim leaf (every ray enters every instance whose box it touches)
        "hw2_davids_camera_000", "hw2_grass_desert",
        "hw3_spheres_dof", "hw3_cornellbox_boxes_dynamic",
        # shading draws random numbers (area light / Roughness > 0): the depth-first pixel loop of rt_ordered.h
        "hw3_cornellbox_area", "hw3_cornellbox_brushed_metal", "hw3_metal_glass_plates",
        # tiles of the author's renders: inline meshes, area light with dielectric / conductor pieces at depth 6,
        # depth of field, 900 spp, and a frame of the tap-water animation (binary PLY, rough material)
        "hw3_chessboard_arealight", "hw3_chessboard_dof_glass_queen", "hw3_deadmau5", "hw3_wine_glass", "hw3_tap_0100"]


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    raytracer_amd.hip_lib()          # fails loudly if the extension is missing
    assert raytracer_amd.hip_lib().rt_device_count() >= 1
    return raytracer_amd


@pytest.mark.parametrize("name", GOLD)
def test_golden_reference_scene(rt, name):
    hs = util.golden_scene(name)
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    tile = util.golden_tile(name) or (0, 0, cam.image_width, cam.image_height)
    rgb, hits = ds.render(cam, *tile, want_hits=True)
    ds.close()
    st = util.image_stats(rgb, util.golden_image(name))
    # the primitive id of the north_star: (kind, object) AND, for meshes, the triangle — the faceIdx this library
    # reports, joined to its vertex ids, must be the face the reference's IntersectTopBVH returned (raytracer.cpp:527-531)
    ok = util.hit_ids_match(hits, util.golden_hits(name), hs)
    assert ok.mean() >= 0.9999, f"hit-id parity {ok.mean()}"
    assert st["within1"] >= 0.999, st
    assert ok.all() and st["exact"] == 1.0, (st, float(ok.mean()))   # the bar this build holds itself to


def _vs_oracle(rt, spec, **opts):
    from oracle import binding as ob
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    for k, v in opts.items():
        ds.set_option(k, v)
    st = rt.Stats()
    rgb, hits = ds.render(cam, want_hits=True, stats=st)
    ds.close()
    o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
    assert (hits == o_hits).all()
    assert (rgb == o_rgb).all(), int((rgb != o_rgb).any(-1).sum())
    assert st.rays_primary == ctr.rays_primary and st.rays_secondary == ctr.rays_secondary and st.rays_shadow == ctr.s_rays
    return st


def test_synthetic_mesh_vs_oracle(rt):
    _vs_oracle(rt, rt.scenes.mesh_scene(grid=96, width=640, height=360))


def test_synthetic_instances_depth6_vs_oracle(rt):
    _vs_oracle(rt, rt.scenes.instanced_scene(grid=64, width=480, height=270))


def test_synthetic_multisample_dof_vs_oracle(rt):
    _vs_oracle(rt, rt.scenes.spheres_scene(200, 150, num_samples=16, dof=True))


@pytest.mark.parametrize("lights", ["1+1", "3+2", "0+3", "1+1 costliest tiles first"])
def test_synthetic_glossy_and_area_light_vs_oracle(rt, lights, monkeypatch):
    """rough reflection / refraction and area lights — the pixel's generator is consumed in
    recursion order (rt_ordered.h).  Image, hit ids and ray counts against the oracle.  A hit posts its light tests
    RT_ORDERED_REQS at a time: "3+2" (three point lights, two area lights) and "0+3" need several batches, with point
    and area lights in one batch and batches that end between two area lights' draws.  The last case forces the
    tile order large renders get (a centre-ray pass, costliest 8x8 tiles handed out first): same bytes, same counts."""
    from oracle import binding as ob
    if lights.endswith("first"):
        monkeypatch.setenv("RT_ORDERED_COST_ORDER", "1")
    spec = rt.scenes.glossy_area_scene(grid=48, width=320, height=200, num_samples=4)
    if lights == "3+2":
        spec.point_lights = spec.point_lights + [((4, 5, 3), (2500, 2600, 2400)), ((0, 7, -3), (1800, 1500, 1500))]
        spec.area_lights = spec.area_lights + [((-2.0, 4.5, 1.0), (0.2, -1.0, 0.1), 1.0, (5000, 4800, 4500))]
    elif lights == "0+3":
        spec.point_lights = []
        spec.area_lights = spec.area_lights + [((-2.0, 4.5, 1.0), (0.2, -1.0, 0.1), 1.0, (5000, 4800, 4500)),
                                               ((2.5, 4.0, -1.5), (-0.3, -1.0, 0.2), 0.8, (4000, 4200, 4600))]
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    st = rt.Stats()
    rgb, hits = ds.render(cam, want_hits=True, stats=st)
    part, _ = ds.render(cam, x0=100, y0=50, w=37, h=21)            # a ragged window of the same frame
    ds.close()
    o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
    assert (hits == o_hits).all()
    assert (rgb == o_rgb).all(), int((rgb != o_rgb).any(-1).sum())
    assert (part == rgb[50:71, 100:137]).all()
    assert st.rays_primary == ctr.rays_primary and st.rays_secondary == ctr.rays_secondary
    # InShadow calls + the FindClosestHit calls of the area-light test (raytracer.cpp:1405)
    assert st.rays_shadow == ctr.s_rays + (ctr.c_rays - ctr.rays_primary - ctr.rays_secondary)


def test_one_launch_and_per_bounce_pipelines_agree(rt):
    """rt_mega.h against rt_kernels.h: all generations of a chunk in one persistent launch (the default up to 2^20
    primary rays) and one launch per bounce trace the same rays and give the same bytes — on synthetic scenes (each
    also against the oracle), on reference scenes, with several chunks, and with a record pool small enough that
    chunks overflow and are redone."""
    from oracle import binding as ob
    cases = [rt.scenes.spheres_scene(200, 150), rt.scenes.mesh_scene(grid=96, width=640, height=360),
             rt.scenes.instanced_scene(grid=64, width=480, height=270), rt.scenes.spheres_scene(200, 150, num_samples=16, dof=True),
             rt.scenes.deep_chain_scene()]
    scenes = [(s.build(), True) for s in cases] + [(util.golden_scene(n), False) for n in ("hw1_cornellbox_recursive", "hw2_metal_glass_plates", "hw3_spheres_dof")]
    for hs, check_oracle in scenes:
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        out = {}
        for flags in (0, rt.RT_FLAG_ONE_LAUNCH, rt.RT_FLAG_PER_BOUNCE_LAUNCHES):
            ds.set_option(rt.RT_OPT_FLAGS, flags | rt.RT_FLAG_TIMING)   # (rt_stats counts launches only with timing on)
            st = rt.Stats()
            rgb, hits = ds.render(cam, want_hits=True, stats=st)
            # (a frame may take several chunks — many lights — and a chunk that overflows its pool is launched again)
            out[flags] = (rgb, hits, (st.rays_primary, st.rays_secondary, st.rays_shadow), st.launches_shadow == 0 and st.launches_closest >= st.chunks)
        one, per = out[rt.RT_FLAG_ONE_LAUNCH], out[rt.RT_FLAG_PER_BOUNCE_LAUNCHES]
        assert one[3] and not per[3]
        # the default: one launch for renders of up to 2^20 primary rays (rt.h: RT_FLAG_ONE_LAUNCH)
        assert out[0][3] == (cam.image_width * cam.image_height * max(cam.num_samples, 1) <= 1 << 20)
        for o in (out[0], one):
            assert (o[0] == per[0]).all() and (o[1] == per[1]).all() and o[2] == per[2]
        # several chunks per frame, and a pool that overflows (chunks are redone with fewer pixels)
        ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_ONE_LAUNCH | rt.RT_FLAG_TIMING)
        ds.set_option(rt.RT_OPT_CHUNK_SAMPLES, 8192)
        st = rt.Stats()
        rgb, hits = ds.render(cam, want_hits=True, stats=st)
        assert st.chunks > 1 and st.launches_closest >= st.chunks and st.launches_shadow == 0
        assert (rgb == per[0]).all() and (hits == per[1]).all() and (st.rays_primary, st.rays_secondary, st.rays_shadow) == per[2]
        ds.close()
        if check_oracle:
            o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
            assert (one[0] == o_rgb).all() and (one[1] == o_hits).all()
            assert one[2] == (ctr.rays_primary, ctr.rays_secondary, ctr.s_rays)
    # pool pressure: 600 000 records hold the shading waves' slabs (256 x 2048) and little else
    hs = rt.scenes.instanced_scene(grid=32, width=320, height=200).build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES)
    a, _ = ds.render(cam)
    ds.close()
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_ONE_LAUNCH | rt.RT_FLAG_TIMING)
    ds.set_option(rt.RT_OPT_POOL_RECORDS, 600000)
    st = rt.Stats()
    b, _ = ds.render(cam, stats=st)
    ds.close()
    assert st.launches_shadow == 0 and st.chunks > 1
    assert (a == b).all()
    # the same on a pool that is re-made over buffers another render has just filled and freed: a chunk that overflows
    # leaves record slots nobody wrote, and what they hold is then stale records of the earlier layout — nothing may be
    # combined from them (k_combine_mega), and a pool sized by a per-bounce render is re-made for the slabs' shadow slots
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES)
    c, _ = ds.render(cam)
    ds.set_option(rt.RT_OPT_POOL_RECORDS, 600000)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_ONE_LAUNCH | rt.RT_FLAG_TIMING)
    st = rt.Stats()
    d, _ = ds.render(cam, stats=st)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES)
    e, _ = ds.render(cam)
    ds.close()
    assert st.launches_shadow == 0 and st.chunks > 1
    assert (c == a).all() and (d == a).all() and (e == a).all()


def test_ragged_sizes_and_tiny_images(rt):
    """Edge tiles (w,h not multiples of 8), 1x1 and 1xN images."""
    from oracle import binding as ob
    for (w, h) in [(1, 1), (1, 17), (13, 1), (37, 29), (65, 9)]:
        spec = rt.scenes.spheres_scene(w, h)
        hs = spec.build()
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        rgb, hits = ds.render(cam, want_hits=True)
        ds.close()
        o_rgb, o_hits, _ = ob.render(hs, cam, want_hits=True)
        assert (rgb == o_rgb).all() and (hits == o_hits).all(), (w, h)


def test_chunking_and_pool_pressure_do_not_change_the_image(rt):
    """Small chunks and a small record pool (forcing the overflow/retry path) give the same bytes."""
    spec = rt.scenes.instanced_scene(grid=32, width=320, height=200)
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    a, _ = ds.render(cam)
    ds.close()
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_POOL_RECORDS, 40000)
    ds.set_option(rt.RT_OPT_CHUNK_SAMPLES, 4096)
    st = rt.Stats()
    b, _ = ds.render(cam, stats=st)
    ds.close()
    assert st.chunks > 4
    assert (a == b).all()


def test_tiles_and_interleaved_rows_stitch_to_the_full_frame(rt):
    """Multi-GPU partition rehearsed on one GPU: windows and interleaved row bands
    reproduce the single-call frame byte for byte (global-pixel seeding)."""
    import torch
    hs = util.golden_scene("hw1_cornellbox_recursive")
    cam = hs.camera(0)
    W, H = cam.image_width, cam.image_height
    ds = rt.DeviceScene(hs, 0)
    full, _ = ds.render(cam)
    tile, _ = ds.render(cam, x0=123, y0=77, w=301, h=211)
    assert (tile == full[77:288, 123:424]).all()
    parts = 3
    merged = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    for p in range(parts):
        fb = torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda:0")
        ds.render_rows_device(cam, 0, 0, W, H, parts, p, fb.data_ptr(), W * 3, stats=rt.Stats())
        rows = rt.multigpu.rows_of(p, parts, H).to("cuda:0")
        merged[rows] = fb[rows]
        other = rt.multigpu.rows_of((p + 1) % parts, parts, H).to("cuda:0")
        assert (fb[other] == 7).all()          # rows of other ranks are left untouched
    ds.close()
    assert (merged.cpu().numpy() == full).all()


def test_gather_rows_assembles_the_frame_of_three_replicas(rt):
    """rt_gather_rows (include/rt.h), the frame assembly of a multi-GPU render, rehearsed with three replicas of the
    scene on one GPU: each renders its interleaved 8-row bands into its own device frame on its own stream, the call
    brings the bands together in replica 0's frame and on the host — byte-equal to a single call's frame; with a frame
    whose height is not a multiple of 8 (a short last band), with padded rows, and device-only (host_rgb NULL)."""
    import numpy as np
    import torch
    for hs, (W, H) in ((util.golden_scene("hw1_cornellbox_recursive"), (800, 800)), (rt.scenes.spheres_scene(203, 149).build(), (203, 149))):
        cam = hs.camera(0)
        ref = rt.DeviceScene(hs, 0)
        full, _ = ref.render(cam)
        ref.close()
        for n, pad in ((3, 0), (2, 40), (5, 0)):
            stride = W * 3 + pad
            scenes = [rt.DeviceScene(hs, 0) for _ in range(n)]
            streams = [torch.cuda.Stream() for _ in range(n)]
            fbs = [torch.full((H, stride), 9 + g, dtype=torch.uint8, device="cuda:0") for g in range(n)]
            for g in range(n):
                scenes[g].render_rows_device(cam, 0, 0, W, H, n, g, fbs[g].data_ptr(), stride, stream=streams[g].cuda_stream)
            host = np.zeros((H, W, 3), dtype=np.uint8)
            rt.DeviceScene.gather_rows(scenes, [f.data_ptr() for f in fbs], stride, W, H, host_rgb=host, streams=[s.cuda_stream for s in streams])
            assert (host == full).all(), (n, pad)
            assert (fbs[0][:, :W * 3].cpu().numpy().reshape(H, W, 3) == full).all()
            if pad:
                assert (fbs[0][:, W * 3:] == 9).all()                # the padding of GPU 0's rows is not written
            # device-only form: the frame is in replica 0's buffer for work enqueued on its stream afterwards
            fbs[0].fill_(1)
            torch.cuda.synchronize()
            scenes[0].render_rows_device(cam, 0, 0, W, H, n, 0, fbs[0].data_ptr(), stride, stream=streams[0].cuda_stream)
            rt.DeviceScene.gather_rows(scenes, [f.data_ptr() for f in fbs], stride, W, H, streams=[s.cuda_stream for s in streams])
            with torch.cuda.stream(streams[0]):
                copy = fbs[0].clone()
            streams[0].synchronize()
            assert (copy[:, :W * 3].cpu().numpy().reshape(H, W, 3) == full).all()
            for s_ in scenes:
                s_.sync()
                s_.close()


def test_full_size_properties_mesh1m(rt):
    """BASELINE configs[1] at full size (1 002 528 triangles, 1920x1080): determinism,
    tile independence, and exact agreement with the oracle on a window."""
    from oracle import binding as ob
    spec = rt.scenes.mesh_scene(709, 1920, 1080)
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_TIMING)              # (rt_stats counts launches only with timing on)
    st = rt.Stats()
    a, ha = ds.render(cam, want_hits=True, stats=st)
    b, hb = ds.render(cam, want_hits=True)
    assert (a == b).all() and (ha == hb).all()                     # idempotent / deterministic
    assert st.rays_primary == 1920 * 1080
    hit = ha[..., 0] != 0xFFFFFFFF
    assert st.rays_shadow == 2 * int(hit.sum())                    # 2 point lights per shaded hit, depth 0
    x0, y0, w, h = 840, 470, 256, 160
    win, hw = ds.render(cam, x0=x0, y0=y0, w=w, h=h, want_hits=True)
    assert st.launches_closest == 1 and st.launches_shadow == 1    # a frame this size: the per-bounce pipeline
    # ... and the one-launch pipeline (rt_mega.h) forced at full size: same bytes, ids and ray counts
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_ONE_LAUNCH | rt.RT_FLAG_TIMING)
    s1 = rt.Stats()
    c, hc = ds.render(cam, want_hits=True, stats=s1)
    ds.close()
    assert s1.launches_closest == 1 and s1.launches_shadow == 0
    assert (c == a).all() and (hc == ha).all()
    assert (s1.rays_primary, s1.rays_secondary, s1.rays_shadow) == (st.rays_primary, st.rays_secondary, st.rays_shadow)
    assert (win == a[y0:y0 + h, x0:x0 + w]).all() and (hw == ha[y0:y0 + h, x0:x0 + w]).all()
    o_rgb, o_hits, _ = ob.render(hs, cam, x0=x0, y0=y0, w=w, h=h, want_hits=True)
    assert (win == o_rgb).all() and (hw == o_hits).all()


@pytest.fixture(scope="module")
def inst10_full(rt):
    """The 10-instance, 10 M-triangle scene of BASELINE configs[2..4], built once for the full-size tests."""
    return rt.scenes.instanced_scene(709, 7680, 4320).build()


def test_full_size_properties_inst10_8k_config4(rt, inst10_full):
    """BASELINE configs[4] at its real size: the 7680x4320 frame of the 10 M-triangle instanced scene, depth 6.
    (a) the whole frame in one call: ray-count identities; (b) a 1024-row band rendered as the 8 interleaved
    row parts of an 8-GPU node, one after the other on this GPU, stitches to the same band rendered in one call,
    byte for byte and id for id, and the parts' ray counts add up to the band's; (c) a window of the band equals
    the oracle's."""
    import torch
    from oracle import binding as ob
    hs = inst10_full
    hs.set_camera_resolution(0, 7680, 4320, 1)
    cam = hs.camera(0)
    W, H = 7680, 4320
    ds = rt.DeviceScene(hs, 0)
    dev = "cuda:0"
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    st = rt.Stats()
    ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, stats=st)
    assert st.rays_primary == W * H and st.chunks == 1
    assert st.rays_secondary > 0 and st.rays_shadow > st.rays_primary          # depth 6: children, two lights per lit hit
    y0, bh = 1664, 1024                                                          # a band through the instances
    band = torch.zeros((bh, W, 3), dtype=torch.uint8, device=dev)
    hits = torch.zeros((bh, W, 2), dtype=torch.int32, device=dev)
    sb = rt.Stats()
    ds.render_device(cam, 0, y0, W, bh, band.data_ptr(), W * 3, d_hits_ptr=hits.data_ptr(), stats=sb)
    assert torch.equal(band, fb[y0:y0 + bh])                                     # a tile does not depend on the rest of the frame
    merged = torch.full((bh, W, 3), 7, dtype=torch.uint8, device=dev)
    mhits = torch.full((bh, W, 2), 7, dtype=torch.int32, device=dev)
    tot = [0, 0, 0]
    for p in range(8):
        part = torch.full((bh, W, 3), 9, dtype=torch.uint8, device=dev)
        phits = torch.full((bh, W, 2), 9, dtype=torch.int32, device=dev)
        sp = rt.Stats()
        ds.render_rows_device(cam, 0, y0, W, bh, 8, p, part.data_ptr(), W * 3, d_hits_ptr=phits.data_ptr(), stats=sp)
        rows = rt.multigpu.rows_of(p, 8, bh).to(dev)
        merged[rows] = part[rows]
        mhits[rows] = phits[rows]
        tot = [tot[0] + sp.rays_primary, tot[1] + sp.rays_secondary, tot[2] + sp.rays_shadow]
    assert torch.equal(merged, band) and torch.equal(mhits, hits)
    assert tot == [sb.rays_primary, sb.rays_secondary, sb.rays_shadow] and sb.rays_primary == W * bh
    ds.close()
    x0, wy, ww, wh = 3712, y0 + 448, 256, 160
    o_rgb, o_hits, _ = ob.render(hs, cam, x0=x0, y0=wy, w=ww, h=wh, want_hits=True)
    assert (band[wy - y0:wy - y0 + wh, x0:x0 + ww].cpu().numpy() == o_rgb).all()
    assert (hits[wy - y0:wy - y0 + wh, x0:x0 + ww].cpu().numpy().view(np.uint32) == o_hits).all()
    assert (o_hits[..., 0] != 0xFFFFFFFF).mean() > 0.3                            # the window looks at geometry


def test_full_size_properties_inst10_4k_64spp_config3(rt, inst10_full):
    """BASELINE configs[3] at its real size, WHOLE frame: 3840x2160 at 64 spp, depth 6 — 530.8 M primary samples in
    eight chunks of the default 64 M, the multi-chunk MT19937 scratch and a 27 GB record pool.  Ray-count identities
    of the frame; a 512-row band rendered on its own, with 16 M-sample chunks, must equal the frame's rows (a pixel
    does not depend on the call's rectangle nor on how it is chunked) and its counts must not depend on the chunking;
    two windows that lie in different chunks of the frame equal the oracle's (bytes and ids)."""
    import torch
    from oracle import binding as ob
    hs = inst10_full
    hs.set_camera_resolution(0, 3840, 2160, 64)
    cam = hs.camera(0)
    assert cam.num_samples == 64 and cam.samples_per_side == 8
    W, H = 3840, 2160
    dev = "cuda:0"
    ds = rt.DeviceScene(hs, 0)
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    hf = torch.zeros((H, W, 2), dtype=torch.int32, device=dev)
    sf = rt.Stats()
    ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, d_hits_ptr=hf.data_ptr(), stats=sf)
    assert sf.chunks >= 8 and sf.rays_primary == W * H * 64
    assert sf.rays_secondary > 0 and sf.rays_shadow > sf.rays_primary            # depth 6: children; two lights per lit hit
    y0, bh = 824, 512
    a = torch.zeros((bh, W, 3), dtype=torch.uint8, device=dev)
    sa = rt.Stats()
    ds.render_device(cam, 0, y0, W, bh, a.data_ptr(), W * 3, stats=sa)
    assert sa.chunks > 1 and sa.rays_primary == W * bh * 64
    ds.set_option(rt.RT_OPT_CHUNK_SAMPLES, 16 << 20)
    b = torch.zeros_like(a)
    sb = rt.Stats()
    ds.render_device(cam, 0, y0, W, bh, b.data_ptr(), W * 3, stats=sb)
    ds.close()
    assert sb.chunks >= 7 and torch.equal(a, b) and torch.equal(a, fb[y0:y0 + bh])
    assert [sa.rays_primary, sa.rays_secondary, sa.rays_shadow] == [sb.rays_primary, sb.rays_secondary, sb.rays_shadow]
    # the frame's chunks are runs of 8-row tile rows of about 270 rows each: the windows lie in the fourth and the sixth
    for (x0, wy, ww, wh) in ((1856, 1048, 64, 40), (1200, 1500, 48, 32)):
        o_rgb, o_hits, _ = ob.render(hs, cam, x0=x0, y0=wy, w=ww, h=wh, want_hits=True)
        assert (fb[wy:wy + wh, x0:x0 + ww].cpu().numpy() == o_rgb).all()
        assert (hf[wy:wy + wh, x0:x0 + ww].cpu().numpy().view(np.uint32) == o_hits).all()
        assert (o_hits[..., 0] != 0xFFFFFFFF).any()
    hs.set_camera_resolution(0, 7680, 4320, 1)


def test_deep_chain_bvh_vs_oracle(rt):
    """a mesh BVH the reference walks with 63 of its 64 stack entries (a 62-level chain,
    scenes.deep_chain_scene) needs ~100 entries of the device's one stack (three pending siblings per two levels, the
    top level and the sentinel on the same stack); the capacity is derived from the BVH depths at upload.  Image,
    hit ids and ray counts against the oracle, untransformed and as a transformed mirror instance."""
    spec = rt.scenes.deep_chain_scene()
    st = _vs_oracle(rt, spec)
    assert st.rays_secondary > 0 and st.rays_shadow > 0


def test_overflow_in_an_asynchronous_render_is_reported(rt):
    """rt_render_tile_device without stats returns before its kernels have run; what they detect must not
    be lost.  With the traversal stack cut to 11 entries (RT_OPT_STACK_CAP, a diagnostic option) the mesh frame
    overflows: the call itself returns RT_OK, rt_sync returns RT_ERR_OVERFLOW — once — and after the option is
    lifted the scene renders the correct frame again."""
    import torch
    hs = rt.scenes.mesh_scene(grid=128, width=320, height=200).build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    good, _ = ds.render(cam)
    fb = torch.zeros((200, 320, 3), dtype=torch.uint8, device="cuda:0")
    ds.set_option(rt.RT_OPT_STACK_CAP, 11)
    ds.render_device(cam, 0, 0, 320, 200, fb.data_ptr(), 320 * 3)                 # asynchronous: no stats, depth 0
    with pytest.raises(rt.RtError, match="stack"):
        ds.sync()
    ds.sync()                                                                      # reported once
    ds.render_device(cam, 0, 0, 320, 200, fb.data_ptr(), 320 * 3)
    with pytest.raises(rt.RtError, match="earlier asynchronous render"):
        ds.render(cam)                                                             # ... or by the next call on the scene
    ds.set_option(rt.RT_OPT_STACK_CAP, 64)
    again, _ = ds.render(cam)
    assert (again == good).all()
    ds.render_device(cam, 0, 0, 320, 200, fb.data_ptr(), 320 * 3)
    ds.sync()
    assert (fb.cpu().numpy() == good).all()
    ds.close()


def test_subtree_donation_is_exact(rt, tmp_path):
    """librt_hip_splittest.so is the same source built so that idle lanes take over subtrees of busy
    rays from the first ray on (in the product that happens only once a generation's cursor is drained).
    Its images and hit ids must equal the product's, which the tests above pin to the reference."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    specs = ["golden:hw2_two_berserkers", "golden:hw1_cornellbox_recursive", "golden:hw2_metal_glass_plates",
             "golden:hw3_cornellbox_boxes_dynamic", "mesh:709:1920:1080", "inst:128:640:360"]
    outs = {}
    for so in ("librt_hip.so", "librt_hip_splittest.so"):
        path = str(tmp_path / (so + ".npz"))
        env = dict(os.environ, RT_HIP_SO=so)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_render_npz.py"), path] + specs,
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[so] = np.load(path)
    a, b = outs["librt_hip.so"], outs["librt_hip_splittest.so"]
    for k in a.files:
        assert (a[k] == b[k]).all(), k
    name = "hw2_two_berserkers"
    assert (b[f"golden:{name}/rgb"] == util.golden_image(name)).all()


def test_errors_are_reported_not_thrown(rt):
    hs = util.golden_scene("hw1_simple")
    ds = rt.DeviceScene(hs, 0)
    with pytest.raises(rt.RtError):
        ds.render(hs.camera(0), x0=-1, y0=0, w=8, h=8)     # negative origin
    cam = util.golden_scene("hw1_simple").camera(0)
    ds2 = rt.DeviceScene(util.golden_scene("hw1_simple"), 0)
    with pytest.raises(rt.RtError):
        ds2.render(cam, x0=700, y0=0, w=200, h=10)         # outside the image
    ds.close()
    ds2.close()


@pytest.mark.gpu
def test_update_scene_keeps_meshes_and_equals_oracle(rt):
    """rt_update_scene: frames of an animation over the same meshes.  Each frame rendered from the
    updated scene must be what the oracle renders for that frame's description (transforms, lights, material,
    camera change; frame 2 adds a top-level sphere; frame 3 needs the depth-first pixel loop)."""
    from oracle import binding as ob
    from raytracer_amd import scenes
    a = scenes.instanced_scene(grid=40, width=192, height=108).build()
    ds = rt.DeviceScene(a, 0)
    for frame in (0, 1, 2, 3, 1):
        hs = a if frame == 0 else scenes.instanced_scene(grid=40, width=192, height=108, frame=frame).build()
        if frame:
            ds.update(hs, verify_meshes=True)
        cam = hs.camera(0)
        rgb, hits = ds.render(cam, want_hits=True)
        o_rgb, o_hits, _ = ob.render(hs, cam, want_hits=True)
        assert (hits == o_hits).all(), frame
        assert (rgb == o_rgb).all(), frame
    # other meshes: refused (sizes), and with the same sizes by the hash
    with pytest.raises(rt.RtError):
        ds.update(scenes.instanced_scene(grid=41, width=192, height=108).build())
    spec = scenes.instanced_scene(grid=40, width=192, height=108)
    spec.meshes[1]["verts"] = spec.meshes[1]["verts"] * np.float32(1.5)
    with pytest.raises(rt.RtError):
        ds.update(spec.build(), verify_meshes=True)
    # the scene is still the last good frame
    hs = scenes.instanced_scene(grid=40, width=192, height=108, frame=1).build()
    rgb, _ = ds.render(hs.camera(0))
    assert (rgb == ob.render(hs, hs.camera(0))[0]).all()
    ds.close()


@pytest.mark.gpu
def test_degenerate_scenes_vs_oracle(rt):
    """Nothing to hit; planes only; no lights; a mesh that is a single triangle (its BVH root is a leaf, no node
    records at all); a scaled instance of it looked at along an axis (rays with zero direction components and an
    infinite object-space bound, the case rt_box.h treats literally)."""
    from raytracer_amd import scenes
    cam = dict(position=(0.0, 0.0, 5.0), gaze_point=(0, 0, 0), up=(0, 1, 0), fovy=40.0, near=1.0, width=65, height=33, num_samples=1)
    mat = dict(ambient=(1, 1, 1), diffuse=(0.7, 0.6, 0.5), specular=(0.4, 0.4, 0.4), phong=20.0)
    mirror = dict(type=rt.ffi.RT_MAT_MIRROR, ambient=(0.1, 0.1, 0.1), diffuse=(0.2, 0.2, 0.2), specular=(1, 1, 1), mirror=(0.8, 0.8, 0.8), phong=50.0)

    def base(name):
        s = scenes.SynthScene(name)
        s.background = (30, 60, 90)
        s.max_depth = 3
        s.vertices = [(0, -1, 0), (-1, -1, -1), (1, -1, -1), (0, 1, -1)]
        s.materials = [dict(mat), dict(mirror)]
        s.camera = dict(cam)
        return s

    cases = []
    s = base("nothing")
    cases.append(s)
    s = base("planes_no_lights")
    s.planes = [dict(point=1, normal=(0, 1, 0), material=1)]
    cases.append(s)
    s = base("plane_and_light")
    s.planes = [dict(point=1, normal=(0, 1, 0), material=2)]
    s.point_lights = [((2, 4, 3), (30000, 30000, 30000))]
    cases.append(s)
    s = base("one_triangle")
    s.point_lights = [((2, 4, 3), (30000, 30000, 30000))]
    s.meshes = [dict(verts=None, faces=np.array([[2, 3, 4]], np.int32), material=1)]
    cases.append(s)
    s = base("scaled_instance_axis_aligned_view")
    s.point_lights = [((0, 0, 6), (30000, 30000, 30000))]
    s.planes = [dict(point=1, normal=(0, 1, 0), material=2)]
    s.meshes = [dict(verts=None, faces=np.array([[2, 3, 4]], np.int32), material=1)]
    s.instances = [dict(base=0, material=1, transform=[("s", (0.25, 0.25, 0.25)), ("t", (0.0, 0.0, 1.0))], reset=False)]
    s.camera = dict(cam, width=33, height=33)      # odd size: the centre pixel looks straight down -z unless jittered
    cases.append(s)
    # one-leaf meshes next to each other in the top level: an untransformed smooth quad and a flat one (both written into the top-level
    # walk as triangles, rt_device.h: RT_INLINE_LEAF_MESH), a transformed copy and a moving one (both entered as meshes), a sphere
    # in front of them and a mirror plane under them — hits, barycentrics (smooth normals) and the order in which equal candidates win
    s = base("one_leaf_meshes_mixed")
    s.vertices += [(-1.5, -0.5, -0.5), (1.5, -0.5, -0.5), (1.5, 1.2, -0.8), (-1.5, 1.2, -0.8), (0.2, 0.1, 0.4)]
    s.point_lights = [((2, 4, 3), (30000, 30000, 30000)), ((-3, 2, 4), (15000, 15000, 20000))]
    s.planes = [dict(point=1, normal=(0, 1, 0), material=2)]
    s.meshes = [dict(verts=None, faces=np.array([[5, 6, 7], [5, 7, 8]], np.int32), material=1, smooth=True),
                dict(verts=None, faces=np.array([[2, 3, 4], [2, 4, 8]], np.int32), material=2),
                dict(verts=None, faces=np.array([[5, 6, 7], [5, 7, 8]], np.int32), material=1, transform=[("t", (0.3, 0.2, 0.6)), ("r", (20.0, 0, 1, 0))])]
    s.instances = [dict(base=0, material=2, transform=[("s", (0.5, 0.5, 0.5)), ("t", (-0.8, 0.3, 1.2))], reset=False)]
    s.spheres = [dict(center=9, radius=0.3, material=1)]
    s.camera = dict(cam, width=97, height=65)
    cases.append(s)
    # a top-level leaf the builder cannot split (40 concentric spheres: equal centroids), next to ordinary leaves: the walk takes
    # however many prims follow a leaf node (the 4-wide records needed a side table above 30)
    s = base("unsplittable_top_level_leaf")
    s.vertices += [(0.3, 0.0, 0.2)]
    s.point_lights = [((2, 4, 3), (30000, 30000, 30000))]
    s.planes = [dict(point=1, normal=(0, 1, 0), material=1)]
    s.materials.append(dict(type=rt.ffi.RT_MAT_DIELECTRIC, diffuse=(0.01, 0.01, 0.01), specular=(0.5, 0.5, 0.5), mirror=(1, 1, 1), phong=100.0,
                            refraction_index=1.4, absorption_coef=(0.02, 0.05, 0.05)))
    s.spheres = [dict(center=5, radius=0.9 - 0.02 * k, material=3 if k % 2 == 0 else 1) for k in range(40)]
    s.meshes = [dict(verts=None, faces=np.array([[2, 3, 4]], np.int32), material=2)]
    s.camera = dict(cam, width=81, height=49)
    cases.append(s)
    for spec in cases:
        _vs_oracle(rt, spec)


@pytest.mark.gpu
def test_device_side_bvh_collapse_equals_host_collapse(rt, tmp_path):
    """rt_upload_scene collapses large mesh BVHs into node records on the device (rt_upload.h: parent links,
    pointer jumping for the depth parity, scan, emit) and small ones on the host; RT_UPLOAD_HOST_COLLAPSE=1 forces
    the host walk for all.  Frames and hit ids must not depend on which one built the records (the full-size
    mesh and a scene that mixes both kinds of BVH)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    specs = ["mesh:709:960:540", "inst:200:640:360", "golden:hw2_two_berserkers"]
    outs = []
    for host in (False, True):
        path = str(tmp_path / f"collapse_{int(host)}.npz")
        env = dict(os.environ)
        env.pop("RT_UPLOAD_HOST_COLLAPSE", None)
        if host:
            env["RT_UPLOAD_HOST_COLLAPSE"] = "1"
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_render_npz.py"), path] + specs,
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    for k in outs[0].files:
        assert (outs[0][k] == outs[1][k]).all(), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLD)
def test_near_first_mode_meets_the_north_star_tolerance_on_reference_scenes(rt, name):
    """RT_FLAG_NEAR_FIRST (include/rt.h) is the one NON-PARITY mode of the library: closest-hit rays visit the
    children of a BVH node nearest first instead of in the reference's left-then-right order (raytracer.cpp:749-753,
    802-806).  It is off by default and not held to byte-exactness, but it must stay inside the north_star's
    tolerance against the reference's own renders: primary hit ids equal on >= 99.99 % of pixels, colours within
    1 LSB on >= 99.9 %."""
    hs = util.golden_scene(name)
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_NEAR_FIRST)
    tile = util.golden_tile(name) or (0, 0, cam.image_width, cam.image_height)
    rgb, hits = ds.render(cam, *tile, want_hits=True)
    ds.close()
    st = util.image_stats(rgb, util.golden_image(name))
    ok = util.hit_ids_match(hits, util.golden_hits(name), hs)      # triangles included
    print(f"near-first {name}: hit ids {ok.mean():.6f} exact {st['exact']:.6f} within1 {st['within1']:.6f}")
    assert ok.mean() >= 0.9999, f"hit-id agreement {ok.mean()}"
    assert st["within1"] >= 0.999, st


@pytest.mark.gpu
def test_near_first_mode_vs_exact_mode_on_synthetic_meshes(rt):
    """Same mode on the bench's scene family at a size the test suite affords (the full-size figures are in
    bench.py's `near_first` object): against the exact mode of the same library, same tolerances; the ray counts
    of the primary generation must agree exactly, and the mode must leave no trace on the next exact render."""
    for spec in (rt.scenes.mesh_scene(grid=256, width=960, height=540),
                 rt.scenes.instanced_scene(grid=128, width=640, height=360),
                 rt.scenes.glossy_area_scene(grid=48, width=320, height=200, num_samples=4)):
        hs = spec.build()
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        s0, s1 = rt.Stats(), rt.Stats()
        rgb0, hits0 = ds.render(cam, want_hits=True, stats=s0)
        ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_NEAR_FIRST)
        rgb1, hits1 = ds.render(cam, want_hits=True, stats=s1)
        ds.set_option(rt.RT_OPT_FLAGS, 0)
        rgb2, hits2 = ds.render(cam, want_hits=True)
        ds.close()
        same = (hits0.reshape(-1, 2) == hits1.reshape(-1, 2)).all(axis=1)
        d = np.abs(rgb0.astype(np.int16) - rgb1.astype(np.int16)).max(axis=-1)
        print(f"near-first {spec.name}: hit ids {same.mean():.6f} exact {(d == 0).mean():.6f} within1 {(d <= 1).mean():.6f}")
        assert same.mean() >= 0.9999 and (d <= 1).mean() >= 0.999
        assert s0.rays_primary == s1.rays_primary
        assert (rgb2 == rgb0).all() and (hits2 == hits0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLD)
def test_rebuilt_mesh_bvh_mode_meets_the_north_star_tolerance_on_reference_scenes(rt, name):
    """rt_upload_scene_ex(RT_UPLOAD_REBUILD_MESH_BVH), the second NON-PARITY mode: the library ignores
    the reference's midpoint-split mesh BVHs (raytracer.cpp:296-339) and builds SAH trees over the same faces
    (rt_rebuild.h); ties go by the caller's leaf order.  Same bar as the near-first mode, against the reference's
    own renders."""
    hs = util.golden_scene(name)
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=True)
    tile = util.golden_tile(name) or (0, 0, cam.image_width, cam.image_height)
    rgb, hits = ds.render(cam, *tile, want_hits=True)
    ds.close()
    st = util.image_stats(rgb, util.golden_image(name))
    ok = util.hit_ids_match(hits, util.golden_hits(name), hs)      # triangles included
    print(f"rebuilt {name}: hit ids {ok.mean():.6f} exact {st['exact']:.6f} within1 {st['within1']:.6f}")
    assert ok.mean() >= 0.9999, f"hit-id agreement {ok.mean()}"
    assert st["within1"] >= 0.999, st


@pytest.mark.gpu
def test_rebuilt_mesh_bvh_mode_vs_exact_mode_and_scene_update(rt):
    """The rebuilt mode against the exact mode of the same library on the bench's scene family, and
    rt_update_scene on a rebuilt scene: the caller passes its own (reference) mesh arrays again, the library keeps
    its own trees."""
    for spec in (rt.scenes.mesh_scene(grid=256, width=960, height=540),
                 rt.scenes.instanced_scene(grid=128, width=640, height=360),
                 rt.scenes.glossy_area_scene(grid=48, width=320, height=200, num_samples=4)):
        hs = spec.build()
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        s0, s1 = rt.Stats(), rt.Stats()
        rgb0, hits0 = ds.render(cam, want_hits=True, stats=s0)
        ds.close()
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=True)
        rgb1, hits1 = ds.render(cam, want_hits=True, stats=s1)
        ds.update(hs, verify_meshes=True)
        rgb2, hits2 = ds.render(cam, want_hits=True)
        ds.close()
        same = (hits0.reshape(-1, 2) == hits1.reshape(-1, 2)).all(axis=1)
        d = np.abs(rgb0.astype(np.int16) - rgb1.astype(np.int16)).max(axis=-1)
        print(f"rebuilt {spec.name}: hit ids {same.mean():.6f} exact {(d == 0).mean():.6f} within1 {(d <= 1).mean():.6f}")
        assert same.mean() >= 0.9999 and (d <= 1).mean() >= 0.999
        assert s0.rays_primary == s1.rays_primary
        assert (rgb2 == rgb1).all() and (hits2 == hits1).all()
    # frames of an animation over a rebuilt scene equal the exact mode's
    frames = [rt.scenes.instanced_scene(grid=64, width=320, height=180, frame=f).build() for f in range(3)]
    ds = rt.DeviceScene(frames[0], 0, rebuild_mesh_bvh=True)
    ex = rt.DeviceScene(frames[0], 0)
    for f, hs in enumerate(frames):
        if f:
            ds.update(hs)
            ex.update(hs)
        a, _ = ds.render(hs.camera(0))
        b, _ = ex.render(hs.camera(0))
        d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=-1)
        assert (d <= 1).mean() >= 0.999, f
    ds.close()
    ex.close()


@pytest.mark.gpu
def test_device_sah_builder_gives_the_host_builders_trees(rt):
    """RT_UPLOAD_REBUILD_ON_DEVICE (rt_dsah.h) restates rt_rebuild.h's binned SAH decision for decision on quantities that do
    not depend on the order of accumulation, so the trees are the host builder's: every frame and every hit id of a scene
    uploaded either way must be EQUAL (ties between equal t go by the caller's leaf order in both).  Scenes: the bench
    family, instances, the reference's scanned mesh and its many-mesh scene, meshes of 1 / 2 / 3 / 64 / 65 / 130 triangles
    (the boundaries between 'one leaf', 'one wave' and 'level by level'), and coincident centroids (halved by position:
    the host halves its own permutation, so there only the tolerance is asked for)."""
    def frames(hs, mode):
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=mode)
        cam = hs.camera(0)
        rgb, hits = ds.render(cam, want_hits=True)
        ds.close()
        return rgb, hits
    specs = [rt.scenes.mesh_scene(grid=200, width=640, height=360), rt.scenes.instanced_scene(grid=96, width=480, height=270),
             rt.scenes.mesh_scene(grid=64, width=320, height=180, tessellation="cube")]
    for spec in specs:
        hs = spec.build()
        a, b = frames(hs, True), frames(hs, "device")
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (spec.name, int((a[0] != b[0]).any(-1).sum()))
    for name in ("hw2_davids_camera_000", "hw2_two_berserkers", "hw1_cornellbox_recursive"):
        hs = util.golden_scene(name)
        a, b = frames(hs, True), frames(hs, "device")
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (name, int((a[0] != b[0]).any(-1).sum()))
    # small counts around the builder's regime boundaries: strips of n triangles side by side
    for n in (1, 2, 3, 5, 63, 64, 65, 66, 129, 130, 257):
        s = rt.scenes.SynthScene(f"strip_{n}")
        s.max_depth = 0
        s.ambient = (30, 30, 30)
        verts, faces = [], []
        for k in range(n):
            x = -2.0 + 4.0 * k / max(n, 2)
            w = 3.6 / max(n, 2)
            verts += [(x, -1.0 + 0.3 * ((k * 7) % 5), -0.2 * (k % 3)), (x + w, -1.0, 0.1 * (k % 4)), (x + 0.5 * w, 1.0 - 0.2 * ((k * 3) % 4), 0.0)]
            faces.append((3 * k, 3 * k + 1, 3 * k + 2))
        s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.7, 0.6, 0.5), specular=(0.2, 0.2, 0.2), phong=10.0)]
        s.point_lights = [((0.5, 3.0, 5.0), (40000, 40000, 40000))]
        s.meshes = [dict(verts=np.array(verts, np.float32), faces=np.array(faces, np.int32), material=1)]
        s.camera = dict(position=(0.0, 0.2, 6.0), gaze_point=(0, 0, 0), up=(0, 1, 0), fovy=45.0, near=1.0, width=256, height=144, num_samples=1)
        hs = s.build()
        ex, a, b, c = frames(hs, False), frames(hs, True), frames(hs, "device"), frames(hs, "lbvh")
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (n, int((a[0] != b[0]).any(-1).sum()))
        for r in (b, c):
            assert (r[1].reshape(-1, 2) == ex[1].reshape(-1, 2)).all(axis=1).mean() >= 0.9999 and (np.abs(r[0].astype(np.int16) - ex[0].astype(np.int16)).max(-1) <= 1).mean() >= 0.999, n
    # coincident centroids: 300 triangles fanned around one centroid plus a normal strip (the reference leaves one big leaf)
    s = rt.scenes.SynthScene("coincident")
    s.max_depth = 0
    s.ambient = (30, 30, 30)
    verts, faces = [], []
    for k in range(300):
        ang = 2.0 * np.pi * k / 300.0
        # three vertices at 120 degrees around the origin of the fan: every centroid is (0, 0, -0.001 k)
        for j in range(3):
            a_ = ang + 2.0 * np.pi * j / 3.0
            verts.append((1.5 * np.cos(a_), 1.5 * np.sin(a_), -0.003 * k))
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    base = len(verts)
    for k in range(80):
        x = -3.0 + 0.07 * k
        verts += [(x, -2.0, 0.5), (x + 0.06, -2.0, 0.5), (x + 0.03, -1.6, 0.5)]
        faces.append((base + 3 * k, base + 3 * k + 1, base + 3 * k + 2))
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.7, 0.6, 0.5), specular=(0.2, 0.2, 0.2), phong=10.0)]
    s.point_lights = [((0.5, 3.0, 5.0), (40000, 40000, 40000))]
    s.meshes = [dict(verts=np.array(verts, np.float32), faces=np.array(faces, np.int32), material=1)]
    s.camera = dict(position=(0.0, 0.0, 6.0), gaze_point=(0, 0, 0), up=(0, 1, 0), fovy=45.0, near=1.0, width=256, height=144, num_samples=1)
    hs = s.build()
    ex = frames(hs, False)
    for mode in (True, "device", "lbvh"):
        r = frames(hs, mode)
        same = (r[1].reshape(-1, 2) == ex[1].reshape(-1, 2)).all(axis=1).mean()
        close = (np.abs(r[0].astype(np.int16) - ex[0].astype(np.int16)).max(-1) <= 1).mean()
        assert same >= 0.9999 and close >= 0.999, (mode, same, close)


@pytest.mark.gpu
def test_mesh_bvhs_rebuilt_on_the_device_meet_the_tolerance(rt):
    """RT_UPLOAD_REBUILD_MESH_BVH | RT_UPLOAD_REBUILD_ON_DEVICE (SURVEY 8f-3 on the GPU): the binned SAH of rt_dsah.h and, with
    RT_UPLOAD_REBUILD_LBVH, linear BVHs built by Morton sort + Karras' radix tree (rt_lbvh.h), both emitted in the reference's node format.  Same bar as the host rebuild: the
    north_star's tolerance against the reference's own renders on mesh scenes of every kind (many overlapping meshes, a
    scanned 100 000-face mesh, instances under a one-leaf top level, inline meshes, a binary PLY with a rough material),
    and against the exact mode on the bench's scene family; rt_update_scene keeps the device-built trees."""
    for name, mode in [(n_, m_) for m_ in ("device", "lbvh") for n_ in ("hw1_cornellbox_recursive", "hw2_two_berserkers", "hw2_davids_camera_000", "hw2_grass_desert",
                                                                      "hw2_metal_glass_plates", "hw3_chessboard_arealight", "hw3_tap_0100")]:
        hs = util.golden_scene(name)
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=mode)
        tile = util.golden_tile(name) or (0, 0, cam.image_width, cam.image_height)
        rgb, hits = ds.render(cam, *tile, want_hits=True)
        ds.close()
        st = util.image_stats(rgb, util.golden_image(name))
        ok = util.hit_ids_match(hits, util.golden_hits(name), hs)
        print(f"device-rebuilt ({mode}) {name}: hit ids {ok.mean():.6f} exact {st['exact']:.6f} within1 {st['within1']:.6f}")
        assert ok.mean() >= 0.9999, (name, ok.mean())
        assert st["within1"] >= 0.999, (name, st)
    for spec in (rt.scenes.mesh_scene(grid=256, width=960, height=540), rt.scenes.instanced_scene(grid=128, width=640, height=360),
                 rt.scenes.deep_chain_scene()):
        hs = spec.build()
        cam = hs.camera(0)
        ds = rt.DeviceScene(hs, 0)
        s0, s1 = rt.Stats(), rt.Stats()
        rgb0, hits0 = ds.render(cam, want_hits=True, stats=s0)
        ds.close()
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh="lbvh" if spec.name.startswith("inst") else "device")
        rgb1, hits1 = ds.render(cam, want_hits=True, stats=s1)
        ds.update(hs, verify_meshes=True)
        rgb2, hits2 = ds.render(cam, want_hits=True)
        ds.close()
        same = (hits0.reshape(-1, 2) == hits1.reshape(-1, 2)).all(axis=1)
        d = np.abs(rgb0.astype(np.int16) - rgb1.astype(np.int16)).max(axis=-1)
        print(f"device-rebuilt {spec.name}: hit ids {same.mean():.6f} exact {(d == 0).mean():.6f}")
        assert same.mean() >= 0.9999 and (d <= 1).mean() >= 0.999
        assert s0.rays_primary == s1.rays_primary
        assert (rgb2 == rgb1).all() and (hits2 == hits1).all()


@pytest.mark.gpu
def test_unsplittable_big_leaves_vs_oracle_in_all_modes(rt):
    """The dragon_new pathology (blog/hw2/episode2.md:240-248): SubdivideMesh gives up when every centroid falls on
    one side of the midpoint (raytracer.cpp:317), leaving a leaf of hundreds of triangles.  Two decks of stacked
    right triangles (centroid at a third of the extent, midpoint at half: never split) of 300 and 100 triangles
    under two interior nodes, one deck a mirror, lit by a point light: closest hits, any-hits and secondary rays
    all run through big leaves — the whole-wave test, the lane-alone test, the side table of leaf sizes.  Exact
    mode against the oracle byte for byte; the two non-parity modes against the exact frame."""
    from oracle import binding as ob
    from raytracer_amd import scenes

    def deck(n, x0, z0, dz, shrink):
        v, f = [], []
        for k in range(n):
            s, z = 1.0 - shrink * k, z0 + dz * k
            v += [(x0, -1.0, z), (x0 + 4.0 * s, -1.0, z), (x0, -1.0 + 2.0 * s, z)]
            f.append((3 * k, 3 * k + 1, 3 * k + 2))
        return v, f

    v1, f1 = deck(300, -5.0, -3.0, 0.002, 0.002)
    v2, f2 = deck(100, 1.0, -2.5, -0.004, 0.005)
    verts = np.array(v1 + v2 + [(-6, -1.2, 2), (6, -1.2, 2), (6, -1.2, -6), (-6, -1.2, -6)], np.float32)
    nb = len(v1) + len(v2)
    faces = np.array(f1 + [(a + len(v1), b + len(v1), c + len(v1)) for a, b, c in f2] + [(nb, nb + 1, nb + 2), (nb, nb + 2, nb + 3)], np.int32)
    s = scenes.SynthScene("big_leaves")
    s.background = (20, 30, 40)
    s.max_depth = 3
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.7, 0.6, 0.5), specular=(0.4, 0.4, 0.4), phong=20.0),
                   dict(type=rt.ffi.RT_MAT_MIRROR, ambient=(0.1, 0.1, 0.1), diffuse=(0.2, 0.2, 0.2), specular=(1, 1, 1), mirror=(0.8, 0.8, 0.8), phong=50.0)]
    s.point_lights = [((1.0, 6.0, 4.0), (60000, 60000, 60000))]
    s.meshes = [dict(verts=verts, faces=faces, material=1),
                dict(verts=np.array(v2, np.float32) + np.float32([0.0, 2.2, 0.3]), faces=np.array(f2, np.int32), material=2)]
    s.camera = dict(position=(0.0, 1.0, 7.0), gaze_point=(0, 0, -2), up=(0, 1, 0), fovy=50.0, near=1.0, width=480, height=270, num_samples=1)
    hs = s.build()
    cam = hs.camera(0)
    # the premise: the reference's builder left big leaves
    import ctypes as C

    class Node(C.Structure):      # rt_bvh_node
        _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left_or_first", C.c_int32), ("prim_count", C.c_int32)]

    class DescHead(C.Structure):  # the first fields of rt_scene_desc (include/rt.h)
        _fields_ = [("background", C.c_float * 3), ("ambient", C.c_float * 3), ("eps", C.c_float), ("max_depth", C.c_int32),
                    ("top_nodes", C.c_void_p), ("n_top_nodes", C.c_uint32), ("top_prim_idx", C.c_void_p), ("n_top_prims", C.c_uint32),
                    ("top_prims", C.c_void_p), ("mesh_bvhs", C.c_void_p), ("n_mesh_bvhs", C.c_uint32),
                    ("mesh_nodes", C.POINTER(Node)), ("n_mesh_nodes", C.c_uint32)]

    d = DescHead.from_address(hs.desc_ptr)
    counts = sorted(d.mesh_nodes[i].prim_count for i in range(d.n_mesh_nodes))
    assert d.n_mesh_bvhs == 2 and counts[-1] >= 300 and counts[-2] >= 100 and counts[-3] >= 100, counts[-4:]
    o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
    ds = rt.DeviceScene(hs, 0)
    st = rt.Stats()
    rgb, hits = ds.render(cam, want_hits=True, stats=st)
    assert (hits == o_hits).all() and (rgb == o_rgb).all(), int((rgb != o_rgb).any(-1).sum())
    assert st.rays_primary == ctr.rays_primary and st.rays_secondary == ctr.rays_secondary and st.rays_shadow == ctr.s_rays
    assert (hits.reshape(-1, 2)[:, 0] != 0xFFFFFFFF).mean() > 0.25         # the decks fill a good part of the frame
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_NEAR_FIRST)
    rgb1, hits1 = ds.render(cam, want_hits=True)
    ds.close()
    ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=True)
    rgb2, hits2 = ds.render(cam, want_hits=True)
    ds.close()
    ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh="device")
    rgb3, hits3 = ds.render(cam, want_hits=True)
    ds.close()
    for name, r, h in (("near-first", rgb1, hits1), ("rebuilt", rgb2, hits2), ("rebuilt on the device", rgb3, hits3)):
        same = (h.reshape(-1, 2) == hits.reshape(-1, 2)).all(axis=1)
        dd = np.abs(r.astype(np.int16) - rgb.astype(np.int16)).max(axis=-1)
        print(f"big leaves, {name}: hit ids {same.mean():.6f} exact {(dd == 0).mean():.6f}")
        assert same.mean() >= 0.9999 and (dd <= 1).mean() >= 0.999
