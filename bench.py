#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X ray path.

Metric (BASELINE.json): Mrays/s, counting every FindClosestHit and InShadow ray.

N = 1   BASELINE.json configs[1] — a 1 002 528-triangle PLY-style mesh (closed-form displaced sphere,
        SURVEY.md §8d M2) at 1920x1080, 1 spp, primary + shadow rays (2 point lights, MaxRecursionDepth 0).
        A "step" is one frame.  The line also carries, as sub-objects that never enter `value`: the two opt-in
        non-parity modes on the same frame, the CPU reference timed on the same frame with a byte-for-byte
        comparison, and `strong_scaling_base` = the N > 1 workload rendered by this one GPU.
N > 1   BASELINE.json configs[4] / SURVEY.md §8d M5 — the FIXED 7680x4320 frame of the 10-instance, 10 M-triangle
        scene (mirror / dielectric / conductor, depth 6), split over the N ranks (strong scaling): rank r renders
        the interleaved 8-row bands r mod N on its own replica of the scene, and the RGB8 rows are gathered to
        rank 0 over RCCL inside the timed region.  `weak` holds the weak-scaling figure of earlier rounds
        (configs[1] with N x the pixels).

  python bench.py --gpus 1 --steps 100 --warmup 5      (the defaults)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Inputs (scene, ray pools) are resident in HBM before the timed region; the frame
stays on the device.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MALL_GATHER_GBS = 8600.0     # same guide, 'Indexed rows': 38 MB table gathered from the Infinity Cache, chip-wide
L2_GATHER_GBS = 17800.0      # same table: rows shared by every workgroup, served by the XCDs' L2 (16.8-18.8 TB/s)
STRONG_W, STRONG_H = 7680, 4320


def weak_frame_size(n_gpus):
    s = math.sqrt(n_gpus)
    return int(round(1920 * s / 8)) * 8, int(round(1080 * s / 8)) * 8


def csrc_sha16():
    """Identity of the kernel sources the library was built from (the GPU box has no .git): profiles/pmc_traffic.json
    records it, and counters taken on other sources are not quoted."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


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


STATS_EVERY = 4      # Run.timed: one step in four is rendered with per-kernel HIP-event times (rt_stats)


class Run:
    """One scene resident on this rank's GPU + its share of one frame; timed() runs W + K steps."""

    def __init__(self, env, spec, rebuild=False, flags=0, hs=None):
        from raytracer_amd import DeviceScene, RT_OPT_FLAGS, RT_FLAG_TIMING
        import torch
        self.env = env
        self.spec = spec
        self.hs = hs if hs is not None else spec.build()
        self.cam = self.hs.camera(0)
        self.W, self.H = self.cam.image_width, self.cam.image_height
        t = time.perf_counter()
        self.ds = DeviceScene(self.hs, env.local_rank, rebuild_mesh_bvh=rebuild)
        self.upload_seconds = time.perf_counter() - t
        self.flags = flags
        self.timing = True
        self.ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | flags)
        self.fb = torch.zeros((self.H, self.W, 3), dtype=torch.uint8, device=env.dev)

    def step(self, st=None):
        from raytracer_amd import multigpu, RT_OPT_FLAGS, RT_FLAG_TIMING
        e = self.env
        if (st is not None) != self.timing:                  # event brackets (and the wait for them) only where rt_stats is asked for
            self.timing = st is not None
            self.ds.set_option(RT_OPT_FLAGS, (RT_FLAG_TIMING if self.timing else 0) | self.flags)
        self.ds.render_rows_device(self.cam, 0, 0, self.W, self.H, e.world, e.rank, self.fb.data_ptr(), self.W * 3, stats=st, stream=e.stream)
        return multigpu.gather_frame(self.fb, e.rank, e.world, force=e.dist)

    def timed(self, steps, warmup):
        import torch
        import torch.distributed as dist
        from raytracer_amd import Stats
        e = self.env
        for _ in range(warmup):
            self.step(Stats())
        acc = {k: 0.0 for k in ("raygen", "closest", "shade", "shadow", "combine", "resolve")}
        acc.update(n_closest=0, n_shadow=0)
        rays = [0, 0, 0]
        e.fence()
        t0 = time.perf_counter()
        frame = None
        # Every STATS_EVERY-th step asks for rt_stats — its kernels are then bracketed by HIP events on the render's stream and
        # the call returns when they are done, which is where the roofline's launch times come from, live, inside the timed
        # region — the others are queued behind each other as a frame loop would queue them (a depth-0 render without
        # rt_stats returns before its kernels have run; include/rt.h).  The sampled times are scaled to all steps.
        sampled = 0
        for i in range(steps):
            st = Stats() if i % STATS_EVERY == 0 else None
            frame = self.step(st)
            if st is None:
                continue
            sampled += 1
            for k in ("raygen", "closest", "shade", "shadow", "combine", "resolve"):
                acc[k] += getattr(st, "ms_" + k)
            acc["n_closest"] += st.launches_closest
            acc["n_shadow"] += st.launches_shadow
            rays = [st.rays_primary, st.rays_secondary, st.rays_shadow]
        e.fence()
        elapsed = time.perf_counter() - t0
        self.ds.sync()                                        # (reports what an asynchronous step left behind, if anything)
        for k in acc:
            acc[k] *= steps / max(sampled, 1)
        t = torch.tensor([elapsed], dtype=torch.float64, device=e.dev)
        r = torch.tensor(rays, dtype=torch.float64, device=e.dev)
        if e.dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the slowest rank's clock
            dist.all_reduce(r, op=dist.ReduceOp.SUM)          # rays of all ranks
        return {"elapsed": float(t.item()), "rays_frame": [float(v) for v in r.tolist()], "rays_rank": rays, "acc": acc,
                "frame": frame, "steps": steps}

    def close(self):
        self.ds.close()


class Env:
    pass


def frame_sha16(frame):
    """Identity of a frame (RGB8 bytes, row-major): equal for every N, since pixels are seeded by their global
    coordinates — the N = 1 line carries it for the strong-scaling workload, every N > 1 line for the gathered frame."""
    return hashlib.sha256(frame.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def share_projection(run, ns=(1, 2, 4, 8), reps=3):
    """One rank's share of the strong-scaling frame on THIS GPU for N = 1, 2, 4, 8 (rank 0's interleaved bands, no
    gather): the time an N-GPU frame cannot beat, i.e. the ceiling of the scaling curve set by the per-launch tails."""
    from raytracer_amd import Stats
    out = {"n": [], "share_ms": [], "speedup_ceiling": []}
    base = None
    for n in ns:
        best = None
        for _ in range(reps):
            st = Stats()
            run.ds.render_rows_device(run.cam, 0, 0, run.W, run.H, n, 0, run.fb.data_ptr(), run.W * 3, stats=st, stream=run.env.stream)
            best = st.ms_total if best is None else min(best, st.ms_total)
        base = best if base is None else base
        out["n"].append(n)
        out["share_ms"].append(best)
        out["speedup_ceiling"].append(base / best)
    return out


def pmc_block(workload_key, kernel):
    """Counters of `kernel` on `workload_key` from the committed rocprofv3 --pmc passes (scripts/pmc.sh ->
    scripts/pmc_to_json.py) — quoted only if they were taken on these very kernel sources."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt.get("csrc_sha16") != csrc_sha16():
            return None
        return pt[workload_key][kernel]
    except (OSError, KeyError, ValueError):
        return None


def roofline_single(acc, rays, wb, workload_key):
    """N = 1, depth 0: one closest-hit and one any-hit launch per frame; the longer one is the dominant kernel."""
    dur_closest = acc["closest"] / max(acc["n_closest"], 1) * 1e-3
    dur_shadow = acc["shadow"] / max(acc["n_shadow"], 1) * 1e-3
    bytes_closest = wb["closest_bytes_per_ray"] * rays[0]
    bytes_shadow = wb["shadow_bytes_per_ray"] * rays[2]
    dom = "k_trace_closest" if dur_closest >= dur_shadow else "k_trace_shadow"
    dom_bytes, dom_dur = (bytes_closest, dur_closest) if dom == "k_trace_closest" else (bytes_shadow, dur_shadow)
    oth_bytes, oth_dur = (bytes_shadow, dur_shadow) if dom == "k_trace_closest" else (bytes_closest, dur_closest)
    pm = pmc_block(workload_key, dom)
    ach = dom_bytes / dom_dur / 1e9
    r = {
        "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
        # bytes per launch at the fabric side of L2, FETCH_SIZE corrected as the microarch guide prescribes
        # (profiles/pmc_traffic.json); null when the committed counters were taken on other kernel sources
        "traffic": pm["hbm_bytes_per_launch"] if pm else None,
        "algorithmic_bytes_per_launch": dom_bytes,
        "algorithmic_bytes_per_ray": wb["closest_bytes_per_ray"] if dom == "k_trace_closest" else wb["shadow_bytes_per_ray"],
        "rays_per_launch": rays[0] if dom == "k_trace_closest" else rays[2],
        "avg_launch_ms": dom_dur * 1e3,
        # The working set (66 MB) lives in L2 / Infinity Cache, so the HBM fraction is not the roof that binds; the
        # same algorithmic bytes against the cache gather rates the guide measured:
        "cache_roofline": {"achieved": ach, "unit": "GB/s", "mall_gather_peak": MALL_GATHER_GBS, "frac_of_mall_gather": ach / MALL_GATHER_GBS,
                           "l2_gather_peak": L2_GATHER_GBS, "frac_of_l2_gather": ach / L2_GATHER_GBS},
        "other_kernel": {"kernel": "k_trace_shadow" if dom == "k_trace_closest" else "k_trace_closest",
                         "achieved": oth_bytes / oth_dur / 1e9 if oth_dur > 0 else None, "avg_launch_ms": oth_dur * 1e3},
    }
    # what binds instead (VALU issue at partial lane occupancy): from the PMC passes, same condition as `traffic`
    for k in ("valu_issue_util", "waves_per_simd", "lane_occupancy", "l2_hit_rate", "valu_insts_per_launch"):
        r[k] = pm.get(k) if pm else None
    # The roof that binds.  The trace loop issues one wave64 VALU instruction per SIMD every 4 cycles at best; what the
    # chip's FP32 lanes deliver is that issue rate times the lanes each instruction has switched on.
    r["binding"] = binding_block(pm)
    # An HBM fraction above 1 is not a measurement error: the bytes are those of the REFERENCE's traversal
    # and a kernel that does less work than that traversal (any-hit exit) moves fewer.  Such a figure ranks nothing.
    r["saturated"] = bool(ach / HBM_PEAK_GBS > 1.0)
    oth = r["other_kernel"]
    if oth["achieved"] is not None:
        oth["frac"] = oth["achieved"] / HBM_PEAK_GBS
        oth["saturated"] = bool(oth["frac"] > 1.0)
        oth["binding"] = binding_block(pmc_block(workload_key, oth["kernel"]))
    return r, bytes_closest


FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 vector peak


def binding_block(pm):
    """VALU issue at partial lane occupancy, from the committed PMC passes (null when taken on other sources)."""
    if not pm or pm.get("valu_issue_util") is None or pm.get("lane_occupancy") is None:
        return None
    useful = pm["valu_issue_util"] * pm["lane_occupancy"]
    b = {"bound": "valu_issue", "achieved": useful * FP32_PEAK_TFLOPS, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s-equivalent lanes",
         "frac": useful, "valu_issue_util": pm["valu_issue_util"], "lane_occupancy": pm["lane_occupancy"],
         "waves_per_simd": pm.get("waves_per_simd")}
    if pm.get("SQ_INSTS_SALU") and pm.get("SQ_INSTS_VALU"):
        b["salu_per_valu"] = pm["SQ_INSTS_SALU"] / pm["SQ_INSTS_VALU"]
        b["salu_insts_per_launch"] = pm["SQ_INSTS_SALU"]
        b["valu_insts_per_launch"] = pm["SQ_INSTS_VALU"]
    if pm.get("trace_avg_ms") is not None:
        b["rocprof_avg_launch_ms"] = pm["trace_avg_ms"]
    return b


def roofline_all_traces(acc, rays_rank, wb, steps):
    """Depth > 0: a frame is a chain of trace launches (k_trace_closest for generation 0, then one k_trace_mixed per
    bounce, a last k_trace_shadow): the figure is over all of them on this rank — algorithmic bytes of the rays
    this rank traced over the summed launch time (bytes per launch over the average launch: the same ratio)."""
    n_launch = acc["n_closest"] + acc["n_shadow"]
    dur = (acc["closest"] + acc["shadow"]) * 1e-3
    frame_bytes = wb["closest_bytes_per_ray"] * (rays_rank[0] + rays_rank[1]) + wb["shadow_bytes_per_ray"] * rays_rank[2]
    return n_launch, dur, frame_bytes * steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=709, help="mesh grid (709 -> 1 002 528 triangles)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rebuilt", action="store_true", help="skip the rebuilt-BVH leg (so that a profile of k_trace_closest<true> is the near-first leg alone)")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the non-parity legs and the strong-scaling base; N > 1: skip the weak-scaling leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import RT_FLAG_NEAR_FIRST, scenes

    e = Env()
    e.rank = int(os.environ.get("RANK", "0"))
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    e.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if e.world != args.gpus and e.world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={e.world}")
    # RT_BENCH_FORCE_DIST=1 (tests/test_gpu_frames.py, never the driver): take the N > 1 branch with whatever world size
    # the launcher gave — with one rank that is a one-rank RCCL communicator, so RCCL initialisation, the barrier, the
    # all-reduces and the gather of uint8 rows all execute on a box with a single GPU.
    e.dist = e.world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Rehearsal on a 1-GPU box: RT_BENCH_REHEARSE=1 puts every rank on GPU 0 and uses gloo (RCCL refuses two
    # ranks on one device).  Never set by the driver; real runs are one rank per GPU over RCCL.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        e.local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)   # a rehearsal that stalls says where
    torch.cuda.set_device(e.local_rank)
    e.dev = torch.device("cuda", e.local_rank)
    if e.dist:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=e.dev)
    e.stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if e.dist:
            dist.barrier()
            torch.cuda.synchronize()
    e.fence = fence

    with open(os.path.join(ROOT, "profiles", "workload_bytes.json")) as f:
        wb_all = json.load(f)
    strong_w, strong_h = (STRONG_W, STRONG_H) if args.grid == 709 else (1920, 1080)   # (--grid: rehearsals at a smaller size)
    line = None

    if not e.dist:
        # ------------------------------------------------------------------ N = 1: configs[1]
        spec = scenes.mesh_scene(args.grid, 1920, 1080)
        run = Run(e, spec)
        res = run.timed(args.steps, args.warmup)
        rays = res["rays_rank"]
        rays_per_frame = sum(res["rays_frame"])
        ms_per_step = res["elapsed"] / args.steps * 1e3
        value = rays_per_frame / (res["elapsed"] / args.steps) / 1e6
        acc = res["acc"]
        exact_frame = res["frame"].clone()
        roofline, bytes_closest = roofline_single(acc, rays, wb_all["mesh1m_1920x1080"], "mesh1m_1920x1080" if args.grid == 709 else "-")
        roofline["stage_ms_per_step"] = {k: acc[k] / args.steps for k in ("raygen", "closest", "shade", "shadow", "combine", "resolve")}
        line = {
            "metric": "Mrays/s (primary+secondary+shadow)", "value": value, "unit": "Mrays/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"mesh1m: {2 * (args.grid - 1) ** 2}-triangle mesh, 1920x1080, 1 spp, primary + shadow rays (2 point lights), depth 0",
                       "rays_per_frame": rays_per_frame, "seconds_per_frame": ms_per_step / 1e3,
                       "partition": "one GPU, whole frame; N > 1 runs BASELINE configs[4] (see strong_scaling_base)"},
            "roofline": roofline,
        }
        if not args.no_extras:
            # The same frames in the non-parity visiting order (RT_FLAG_NEAR_FIRST, include/rt.h) — reported beside
            # the headline, never as `value`; compared with the exact frame pixel by pixel.
            def extra_leg(r, note):
                x = r.timed(args.steps, 1)
                dmax = (x["frame"].to(torch.int16) - exact_frame.to(torch.int16)).abs().amax(dim=2)
                ms_c = x["acc"]["closest"] / args.steps
                out = {"value": rays_per_frame / (x["elapsed"] / args.steps) / 1e6, "unit": "Mrays/s", "ms_per_step": x["elapsed"] / args.steps * 1e3,
                       "closest_ms_per_step": ms_c, "shadow_ms_per_step": x["acc"]["shadow"] / args.steps,
                       "pixels_equal_to_exact_frame": float((dmax == 0).float().mean().item()),
                       "pixels_within_1_lsb": float((dmax <= 1).float().mean().item()), "note": note}
                if ms_c > 0:   # the same algorithmic bytes (the reference's traversal) over this mode's closest-hit launches
                    out["closest_achieved_GBps"] = bytes_closest / (ms_c * 1e-3) / 1e9
                    out["closest_frac_of_hbm_peak"] = out["closest_achieved_GBps"] / HBM_PEAK_GBS
                return out
            r2 = Run(e, spec, flags=RT_FLAG_NEAR_FIRST, hs=run.hs)
            line["near_first"] = extra_leg(r2, "non-parity mode: children of a BVH node visited nearest first; opt-in (RT_FLAG_NEAR_FIRST)")
            r2.close()
            if not args.no_rebuilt:
                r3 = Run(e, spec, rebuild=True, hs=run.hs)
                line["rebuilt_bvh"] = extra_leg(r3, "non-parity mode: binned-SAH mesh BVHs built by the library, traced nearest first; opt-in (rt_upload_scene_ex)")
                line["rebuilt_bvh"]["upload_seconds"] = r3.upload_seconds
                r3.close()
                r4 = Run(e, spec, rebuild="device", hs=run.hs)
                line["rebuilt_bvh_on_device"] = extra_leg(r4, "non-parity mode: the same binned-SAH trees built on the GPU (rt_dsah.h), traced nearest first; opt-in (rt_upload_scene_ex)")
                line["rebuilt_bvh_on_device"]["upload_seconds"] = r4.upload_seconds
                r4.close()
                r5 = Run(e, spec, rebuild="lbvh", hs=run.hs)
                line["rebuilt_lbvh_on_device"] = extra_leg(r5, "non-parity mode: linear BVHs built on the GPU (Morton sort + radix tree, rt_lbvh.h), traced nearest first; opt-in (rt_upload_scene_ex)")
                line["rebuilt_lbvh_on_device"]["upload_seconds"] = r5.upload_seconds
                r5.close()
            # the workload of the N > 1 runs on this one GPU: the base of the strong-scaling curve
            rs = Run(e, scenes.instanced_scene(args.grid, strong_w, strong_h))
            k = max(2, min(5, args.steps))
            x = rs.timed(k, 1)
            n_launch, dur, bts = roofline_all_traces(x["acc"], x["rays_rank"], wb_all["inst10_7680x4320"], k)
            line["strong_scaling_base"] = {
                "workload": f"inst10: 10 instances of the {2 * (args.grid - 1) ** 2}-triangle mesh, {strong_w}x{strong_h}, 1 spp, depth 6 (BASELINE configs[4])",
                "value": sum(x["rays_frame"]) / (x["elapsed"] / k) / 1e6, "unit": "Mrays/s", "n_gpus": 1, "steps": k,
                "ms_per_step": x["elapsed"] / k * 1e3, "seconds_per_frame": x["elapsed"] / k, "rays_per_frame": sum(x["rays_frame"]),
                "trace_launches_per_frame": n_launch / k,
                "trace_achieved_GBps": bts / dur / 1e9 if dur > 0 else None, "trace_frac_of_hbm_peak": bts / dur / 1e9 / HBM_PEAK_GBS if dur > 0 else None,
                "frame_sha16": frame_sha16(x["frame"])}
            # one rank's share of that frame for N = 1, 2, 4, 8 on this GPU: what the per-launch tails leave of ideal scaling
            line["strong_scaling_projection"] = share_projection(rs)
            rs.close()
        if not args.no_cpu:
            cb, parity = cpu_baseline(spec, exact_frame.cpu().numpy())
            line["cpu_baseline"] = cb
            if parity:
                line["parity_vs_cpu_reference"] = parity
        run.close()
    else:
        # ------------------------------------------------------------------ N > 1: configs[4], strong scaling
        spec = scenes.instanced_scene(args.grid, strong_w, strong_h)
        run = Run(e, spec)
        res = run.timed(args.steps, args.warmup)
        rays_per_frame = sum(res["rays_frame"])
        ms_per_step = res["elapsed"] / args.steps * 1e3
        value = rays_per_frame / (res["elapsed"] / args.steps) / 1e6
        n_launch, dur, bts = roofline_all_traces(res["acc"], res["rays_rank"], wb_all["inst10_7680x4320"], args.steps)
        sha = frame_sha16(res["frame"]) if e.rank == 0 else None       # the gathered frame (equal to the one-GPU frame's)
        run.close()
        weak = None
        if not args.no_extras:
            W, H = weak_frame_size(e.world)
            rw = Run(e, scenes.mesh_scene(args.grid, W, H))
            k = max(2, min(10, args.steps))
            x = rw.timed(k, 1)
            weak = {"workload": f"mesh1m at {W}x{H} (configs[1] with {e.world} x the pixels)", "scaling": "weak",
                    "value": sum(x["rays_frame"]) / (x["elapsed"] / k) / 1e6, "unit": "Mrays/s", "ms_per_step": x["elapsed"] / k * 1e3, "steps": k}
            rw.close()
        if e.rank == 0:
            ach = bts / dur / 1e9 if dur > 0 else 0.0
            line = {
                "metric": "Mrays/s (primary+secondary+shadow)", "value": value, "unit": "Mrays/s", "n_gpus": e.world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"inst10: 10 instances of the {2 * (args.grid - 1) ** 2}-triangle mesh (mirror / dielectric / conductor), {strong_w}x{strong_h}, 1 spp, depth 6 (BASELINE configs[4])",
                           "rays_per_frame": rays_per_frame, "seconds_per_frame": ms_per_step / 1e3,
                           "partition": f"fixed frame; interleaved 8-row bands over {e.world} GPUs, scene replicated, RGB8 rows gathered to rank 0 (one gather, {strong_w * strong_h * 3} bytes in all)",
                           "single_gpu_figure": "the N = 1 line's strong_scaling_base is this workload on one GPU",
                           "frame_sha16": sha, "gather": "gloo (rehearsal)" if rehearse else "rccl"},
                "roofline": {"bound": "hbm", "kernel": "all trace launches of rank 0 (k_trace_closest, k_trace_mixed per bounce, k_trace_shadow)",
                             "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                             "launches_per_frame": n_launch / args.steps, "avg_launch_ms": dur * 1e3 / max(n_launch, 1),
                             "algorithmic_bytes_per_launch": bts / max(n_launch, 1),
                             "cache_roofline": {"achieved": ach, "unit": "GB/s", "mall_gather_peak": MALL_GATHER_GBS, "frac_of_mall_gather": ach / MALL_GATHER_GBS},
                             "stage_ms_per_step": {k: res["acc"][k] / args.steps for k in ("raygen", "closest", "shade", "shadow", "combine", "resolve")}},
            }
            if weak:
                line["weak"] = weak
    if e.rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if e.dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
