#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray path.

Metric (BASELINE.json): Mrays/s, counting every FindClosestHit and InShadow ray.
Workload at N=1: BASELINE.json configs[1] — a 1 002 528-triangle PLY-style mesh
(closed-form displaced sphere) at 1920x1080, 1 spp, primary +
shadow rays (2 point lights, MaxRecursionDepth 0).  A "step" is one frame.
At N>1 the frame keeps the same view and grows to N x the pixels (weak scaling);
rank r renders the interleaved 8-row bands r mod N on its own scene replica and
the RGB8 rows are gathered to rank 0 over RCCL inside the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Inputs (scene, ray pools) are resident in HBM
before the timed region; the frame stays on the device.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def frame_size(n_gpus):
    s = math.sqrt(n_gpus)
    return int(round(1920 * s / 8)) * 8, int(round(1080 * s / 8)) * 8


def cpu_baseline(scene_spec, gpu_rgb):
    """Time the reference itself (oracle/_ref, compiled from the reference's own
    sources) — or, if that library was not built, the C restatement — on the host
    cores, on the SAME full frame, and compare the GPU frame with it."""
    from oracle import binding as ob
    cam_w, cam_h = scene_spec.camera["width"], scene_spec.camera["height"]
    out = {"unit": "Mrays/s"}
    hs = scene_spec.build()
    cam = hs.camera(0)
    t0 = time.time()
    rgb_port, _, ctr = ob.render(hs, cam)          # also yields the ray counts (identical ray set)
    t_port = time.time() - t0
    rays = ctr.c_rays + ctr.s_rays
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    if ob.have_ref():
        with tempfile.TemporaryDirectory() as td:
            js = scene_spec.write_json(td)
            ob.ref_load(js)
            threads = ob.ref().ref_omp_max_threads()
            rgb_ref, sec = ob.ref_render(0, 0, 0, cam_w, cam_h)
        out.update(value=rays / sec / 1e6, kind="reference", cores=threads, seconds=sec,
                   sample=f"whole frame {cam_w}x{cam_h}, {rays} rays, reference functions under its OpenMP loop")
        out["port_seconds"] = t_port
        ref_img = rgb_ref
        out["port_equals_reference"] = bool((rgb_port == rgb_ref).all())
    else:
        out.update(value=rays / t_port / 1e6, kind="port", cores=threads, seconds=t_port,
                   sample=f"whole frame {cam_w}x{cam_h}, {rays} rays, C restatement (oracle/rt_oracle.c)")
        ref_img = rgb_port
    parity = None
    if gpu_rgb is not None:
        import numpy as np
        d = np.abs(gpu_rgb.astype(np.int32) - ref_img.astype(np.int32)).max(-1)
        parity = {"pixels": int(d.size), "byte_exact_fraction": float((d == 0).mean()),
                  "within_1lsb_fraction": float((d <= 1).mean()), "max_abs_diff": int(d.max())}
    return out, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=709, help="mesh grid (709 -> 1 002 528 triangles)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rebuilt", action="store_true", help="skip the rebuilt-BVH leg (so that a profile of k_trace_closest<true> is the near-first leg alone)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, RT_FLAG_NEAR_FIRST, scenes, multigpu

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Rehearsal on a 1-GPU box: RT_BENCH_REHEARSE=1 puts every rank on GPU 0 and uses gloo (RCCL refuses two
    # ranks on one device).  Never set by the driver; real runs are one rank per GPU over RCCL.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H = frame_size(world)
    spec = scenes.mesh_scene(args.grid, W, H)
    hs = spec.build()
    cam = hs.camera(0)
    ds = DeviceScene(hs, local_rank)
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(st=None):
        ds.render_rows_device(cam, 0, 0, W, H, world, rank, fb.data_ptr(), W * 3, stats=st, stream=stream)
        return multigpu.gather_frame(fb, rank, world)

    for _ in range(args.warmup):
        step(Stats())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    acc = {"closest": 0.0, "shadow": 0.0, "n_closest": 0, "n_shadow": 0, "raygen": 0.0, "shade": 0.0, "combine": 0.0, "resolve": 0.0}
    rays = [0, 0, 0]
    fence()
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        st = Stats()
        frame = step(st)
        acc["closest"] += st.ms_closest; acc["shadow"] += st.ms_shadow
        acc["n_closest"] += st.launches_closest; acc["n_shadow"] += st.launches_shadow
        acc["raygen"] += st.ms_raygen; acc["shade"] += st.ms_shade; acc["combine"] += st.ms_combine; acc["resolve"] += st.ms_resolve
        rays = [st.rays_primary, st.rays_secondary, st.rays_shadow]
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    r = torch.tensor(rays, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_per_frame = float(r.sum().item())

    # The same frames in the non-parity visiting order (RT_FLAG_NEAR_FIRST, include/rt.h) — reported beside the
    # headline, never as `value`; compared with the exact frame pixel by pixel.
    near = rebuilt = None
    if world == 1:
        exact_frame = frame.clone()
        ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | RT_FLAG_NEAR_FIRST)
        step(Stats())
        fence()
        t1 = time.perf_counter()
        ms_c = 0.0
        for _ in range(args.steps):
            st = Stats()
            nf = step(st)
            ms_c += st.ms_closest
        fence()
        e1 = time.perf_counter() - t1
        dmax = (nf.to(torch.int16) - exact_frame.to(torch.int16)).abs().amax(dim=2)
        near = {"value": rays_per_frame / (e1 / args.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": e1 / args.steps * 1e3,
                "closest_ms_per_step": ms_c / args.steps,
                "pixels_equal_to_exact_frame": float((dmax == 0).float().mean().item()),
                "pixels_within_1_lsb": float((dmax <= 1).float().mean().item()),
                "note": "non-parity mode: children of a BVH node visited nearest first; opt-in (RT_FLAG_NEAR_FIRST)"}
        ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
        frame = exact_frame           # (fb now holds the near-first frame; the parity check is on the exact one)
    if world == 1 and not args.no_rebuilt:
        # ... and with the library's own SAH trees over the same faces (rt_upload_scene_ex, RT_UPLOAD_REBUILD_MESH_BVH)
        t2 = time.perf_counter()
        ds2 = DeviceScene(hs, local_rank, rebuild_mesh_bvh=True)
        upload_s = time.perf_counter() - t2
        ds2.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
        fb2 = torch.zeros_like(fb)

        def step2(st):
            ds2.render_rows_device(cam, 0, 0, W, H, world, rank, fb2.data_ptr(), W * 3, stats=st, stream=stream)

        step2(Stats())
        fence()
        t2 = time.perf_counter()
        ms_c = ms_s = 0.0
        for _ in range(args.steps):
            st = Stats()
            step2(st)
            ms_c += st.ms_closest; ms_s += st.ms_shadow
        fence()
        e2 = time.perf_counter() - t2
        dmax = (fb2.to(torch.int16) - exact_frame.to(torch.int16)).abs().amax(dim=2)
        rebuilt = {"value": rays_per_frame / (e2 / args.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": e2 / args.steps * 1e3,
                   "closest_ms_per_step": ms_c / args.steps, "shadow_ms_per_step": ms_s / args.steps,
                   "pixels_equal_to_exact_frame": float((dmax == 0).float().mean().item()),
                   "pixels_within_1_lsb": float((dmax <= 1).float().mean().item()),
                   "upload_seconds": upload_s,
                   "note": "non-parity mode: binned-SAH mesh BVHs built by the library, traced nearest first; opt-in (rt_upload_scene_ex)"}
        ds2.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame / (elapsed / args.steps) / 1e6
        with open(os.path.join(ROOT, "profiles", "workload_bytes.json")) as f:
            wb = json.load(f)["mesh1m_1920x1080"]
        # dominant kernel: the closest-hit traversal (one launch per frame at depth 0)
        dur_closest = acc["closest"] / max(acc["n_closest"], 1) * 1e-3
        dur_shadow = acc["shadow"] / max(acc["n_shadow"], 1) * 1e-3
        bytes_closest = wb["closest_bytes_per_ray"] * rays[0]
        bytes_shadow = wb["shadow_bytes_per_ray"] * rays[2]
        dom = "k_trace_closest" if dur_closest >= dur_shadow else "k_trace_shadow"
        # HBM-side bytes per launch of that kernel, from the committed rocprofv3 --pmc passes of this same
        # command (a bench run cannot collect counters itself); only quoted for the workload they were taken on
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if world == 1 and args.grid == 709:
                traffic = pt["mesh1m_1920x1080"][dom]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        ach = (bytes_closest / dur_closest if dom == "k_trace_closest" else bytes_shadow / dur_shadow) / 1e9
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,   # bytes per launch, FETCH_SIZE corrected as the microarch guide prescribes (profiles/pmc_traffic.json)
            "algorithmic_bytes_per_launch": bytes_closest if dom == "k_trace_closest" else bytes_shadow,
            "algorithmic_bytes_per_ray": wb["closest_bytes_per_ray"] if dom == "k_trace_closest" else wb["shadow_bytes_per_ray"],
            "rays_per_launch": rays[0] if dom == "k_trace_closest" else rays[2],
            "avg_launch_ms": (dur_closest if dom == "k_trace_closest" else dur_shadow) * 1e3,
            "other_kernel": {"kernel": "k_trace_shadow" if dom == "k_trace_closest" else "k_trace_closest",
                             "achieved": (bytes_shadow / dur_shadow if dom == "k_trace_closest" else bytes_closest / dur_closest) / 1e9 if min(dur_closest, dur_shadow) > 0 else None,
                             "avg_launch_ms": (dur_shadow if dom == "k_trace_closest" else dur_closest) * 1e3},
            "stage_ms_per_step": {k: acc[k] / args.steps for k in ("raygen", "closest", "shade", "shadow", "combine", "resolve")},
        }
        line = {
            "metric": "Mrays/s (primary+secondary+shadow)", "value": value, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mesh1m: {2 * (args.grid - 1) ** 2}-triangle mesh, {W}x{H}, 1 spp, primary + shadow rays (2 point lights), depth 0",
                       "rays_per_frame": rays_per_frame, "seconds_per_frame": ms_per_step / 1e3,
                       "partition": f"interleaved 8-row bands over {world} GPU(s), scene replicated, RGB8 rows gathered to rank 0"},
            "roofline": roofline,
        }
        # the same algorithmic bytes (the reference's traversal) over these modes' closest-hit launches
        for extra in (near, rebuilt):
            if extra is not None and extra["closest_ms_per_step"] > 0:
                extra["closest_achieved_GBps"] = bytes_closest / (extra["closest_ms_per_step"] * 1e-3) / 1e9
                extra["closest_frac_of_hbm_peak"] = extra["closest_achieved_GBps"] / HBM_PEAK_GBS
        if near is not None:
            line["near_first"] = near
        if rebuilt is not None:
            line["rebuilt_bvh"] = rebuilt
        if world == 1 and not args.no_cpu:
            cb, parity = cpu_baseline(spec, frame.cpu().numpy() if frame is not None else None)
            line["cpu_baseline"] = cb
            if parity:
                line["parity_vs_cpu_reference"] = parity
        print(json.dumps(line), flush=True)
    ds.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
