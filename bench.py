#!/usr/bin/env python3
"""bench.py -- Mray/s of the ndt ray-trace hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one render_image call (reference ndt.c:900) of the workload frame.

Workload (N=1): BASELINE.json configs[2] -- scenes/random.c, 4-D, 1920x1080, `-l 4`, full
object-plugin set + kd-tree -- the configuration the north_star's target is quoted on.  The
scene is the committed fixture tests/golden/c3_random4d.ndtscene.gz (flattened from the
compiled reference), i.e. synthetic data.
N>1: BASELINE.json configs[3] -- the same scene and camera at 3840x2160 for EVERY N > 1, rows
dealt cyclically to the ranks like the reference's MPI_ROW mode (ndt.c:812-820), one RCCL
gather over xGMI assembling the final 8-bit image on rank 0 (--gather f64: the double
framebuffer): total work is fixed as N grows, "scaling": "strong"; rank 0 also times the whole
3840x2160 frame on its one GPU after the run (`strong_reference`), the time the N-GPU frame is
to be compared with.  `--scaling weak` keeps per-GPU work fixed instead (N x 1080p pixels).

`value` counts rays ACTUALLY traced on the GPUs (one trace_kd query each).  The reference
re-traces every pixel's identical ray tree k = 3..18 times (adaptive loop, ndt.c:488; SURVEY
8a row A3); the GPU path traces it once and replays the loop's arithmetic, so the
reference-equivalent rate (what the CPU baseline's rays/time means) is reported separately as
`mray_s_ref_equiv`, never as `value`.
"""
import argparse
import ctypes as C
import json
import math
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libndt_hip: one HIP runtime per process, see ndt_amd/hip.py)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (fixture, depth, reference scene .so, dims, BASELINE config index)
    "random4d": ("c3_random4d", 4, "random", 4, 2),
    "balls4d": ("c2_balls4d", 128, "balls", 4, 1),
    "hypercube3d": ("c1_hypercube3d", 128, "hypercube", 3, 0),
    # configs[4]: the 6-D .. 8-D sweep (728 / 2186 / 6560 objects: scene in global memory, visit masks in a slab)
    "hypercube6d": ("c5_hypercube6d", 128, "hypercube", 6, 4),
    "hypercube7d": ("c5_hypercube7d", 128, "hypercube", 7, 4),
    "hypercube8d": ("c5_hypercube8d", 128, "hypercube", 8, 4),
}


def frame_size(n_gpus, scaling="strong"):
    """strong: configs[2]'s 1920x1080 on one GPU, configs[3]'s 3840x2160 on any N > 1.
    weak: N x 1080p pixels at 16:9, multiples of 8 (N=1: 1920x1080, N=4: 3840x2160)."""
    if scaling == "strong":
        return (1920, 1080) if n_gpus == 1 else (3840, 2160)
    s = math.sqrt(n_gpus)
    w = int(round(1920 * s / 8.0)) * 8
    h = int(round(1080 * s / 8.0)) * 8
    return w, h


def algorithmic_bytes_per_ray(dims):
    """ray record w+r, hit record w+r, framebuffer RMW = 64N + 168 bytes."""
    return 64 * dims + 168


def f64_frames_in_flight(torch, height, width):
    return torch.zeros((height, width, 4), dtype=torch.float64, device="cuda")


def committed_profile(workload, width, height):
    """The committed rocprofv3 session of this very command (profiles/profile_workload.sh -> profiles/make_profile_json.py):
    HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE passes (x2 on FETCH_SIZE per the gfx950
    rule, checked on k_finish_pixels), its FP64-issue figures, and the hash of the library it was measured on.  Counters
    cannot be read from inside a run; a profile of ANOTHER build is not quoted (traffic stays null, "profile_stale")."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "r04_profile_%s_%dp.json" % (workload, height))
    if width != 1920 or not os.path.exists(path):
        return None, None
    with open(path) as f:
        prof = json.load(f)
    lib = os.path.join(ROOT, "ndt_amd", "libndt_hip.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    return prof, prof.get("lib_sha16") == sha


def cpu_baseline(workload, width, height, depth, threads=16):
    """Time the reference itself (oracle/_ref, built from /root/reference by oracle/Makefile) on
    the host cores of this box, for the same frame.  Falls back to the oracle port if the
    reference build did not travel.  threads=1: the reference's single-thread rate (SURVEY 8d asks for both), on a
    quarter-size frame (half the width, half the height: about a quarter of the rays) so that it stays a bounded sample."""
    fixture, _, scene_so, dims, _ = WORKLOADS[workload]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box owns a 16-core share of the host; the reference's pthread path also stops
    # scaling there (per-ray calloc/free + row imbalance: 256 threads are 2x SLOWER than 16)
    cores = min(avail, threads)
    if threads == 1:
        width, height = width // 2, height // 2
    shim = os.path.join(ROOT, "oracle", "_ref", "ndt_ref_shim")
    if os.path.exists(shim):
        cmd = [shim, "--objects", os.path.join(ROOT, "oracle", "_ref", "objects"),
               "--scene", os.path.join(ROOT, "oracle", "_ref", "scenes", scene_so + ".so"),
               "--dims", str(dims), "--res", "%dx%d" % (width, height), "--threads", str(cores),
               "--depth", str(depth), "--tmp", "/tmp"]
        try:
            out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600).stdout
            sec = float(re.search(r"ref_shim: render_s ([0-9.]+)", out).group(1))
            rays = int(re.search(r"rays_total (\d+)", out).group(1))
            return {"value": rays / sec / 1e6, "unit": "Mray/s", "cores": cores, "kind": "reference",
                    "sample": "%s %d-D %dx%d -l %d, whole frame once, reference render_image with %d pthreads "
                              "(%.2f s, %d trace_kd calls)" % (scene_so, dims, width, height, depth, cores, sec, rays),
                    "frame_s": sec, "rays": rays}
        except Exception as e:  # fall through to the port, loudly: `kind` is what the reader trusts
            sys.stderr.write("cpu_baseline: WARNING: the compiled reference failed (%s); timing the oracle PORT instead (kind: \"port\")\n" % e)
    else:
        sys.stderr.write("cpu_baseline: WARNING: oracle/_ref/ndt_ref_shim is missing (build it with `make -C oracle ref` where /root/reference "
                         "exists); timing the oracle PORT instead (kind: \"port\")\n")
    from ndt_amd import load_scene, RenderParams, RenderStats
    so = os.path.join(ROOT, "oracle", "libndt_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True)
    lib = C.CDLL(so)
    fs = load_scene(os.path.join(ROOT, "tests", "golden", fixture + ".ndtscene.gz"))
    # literal re-sampling (flags=1): the port then does the reference's full work, k samples per pixel
    sh = max(8, height // 4)
    p = RenderParams(width, height, depth, 1, 0, max(1, height // sh), 1, 0)
    st = RenderStats()
    rows = (height + p.row_step - 1) // p.row_step
    buf = np.zeros((rows, width, 4))
    t0 = time.time()
    lib.ndt_oracle_render(fs.byref(), C.byref(p), buf.ctypes.data_as(C.c_void_p), C.byref(st), C.c_int(cores), C.c_int(1))
    sec = time.time() - t0
    return {"value": st.rays_ref_equiv / sec / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "degraded": "the compiled reference (oracle/_ref) was not available on this box: this is the oracle port's time",
            "sample": "%s %dx%d -l %d, every %d-th row, oracle port with literal re-sampling, %d threads" % (
                fixture, width, height, depth, p.row_step, cores),
            "frame_s": sec * p.row_step, "rays": int(st.rays_ref_equiv)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="random4d", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="pthreads for the reference CPU baseline")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = BASELINE configs[3]'s 3840x2160 frame at every N (total work fixed); weak = N x 1080p "
                         "pixels (per-GPU work fixed)")
    ap.add_argument("--gather", default="rgba8", choices=["f64", "rgba8"],
                    help="what the image gather moves: rgba8 = the final 8-bit image (pixel_d2c on the device, what the "
                         "reference writes to disk), f64 = the double framebuffer (8x the bytes over xGMI)")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="every M-th timed step carries the HIP events that time the trace launches (the events cost ~12 us of "
                         "stream time per launch: 3 %% of a frame when every step has them); 1 = every step")
    ap.add_argument("--selftest-gather", action="store_true",
                    help="--gpus 1 only: run the N>1 frame loop (double-buffered RCCL gather, events, de-interleave) in a world "
                         "of one, to exercise that code path on a single GPU; with --verify the frame and gather buffers "
                         "are poisoned before every frame and the last gathered frame is checked")
    ap.add_argument("--verify", action="store_true",
                    help="after the run, rank 0 renders the whole frame alone and compares it with the last gathered one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo = rehearsal of the N>1 code path on a "
                         "one-GPU box (all ranks share device 0, shards staged through host memory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libndt_hip has no CPU path")
    rehearsal = world > 1 and args.backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.selftest_gather        # the gather path runs
    if multi:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        elif rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ndt_amd import load_scene
    from ndt_amd.hip import NdtHip

    fixture, depth, _, dims, cfg_idx = WORKLOADS[args.workload]
    fs = load_scene(os.path.join(ROOT, "tests", "golden", fixture + ".ndtscene.gz"))
    width, height = frame_size(world, args.scaling)
    if world > 1 and args.scaling == "strong":
        cfg_idx = 3 if args.workload == "random4d" else cfg_idx
    gpu = NdtHip(local_rank)
    gpu.upload_scene(fs)

    from ndt_amd.multi import RowGather
    gdev = "cpu" if rehearsal else "cuda"
    # Two gather buffers: the gather of frame k runs (RCCL's own stream) while frame k+1 is rendered, as an
    # animation would be produced; every frame's gather is complete before the timed region ends (fence()).
    n_buf = 2 if multi else 1
    gdtype = torch.uint8 if args.gather == "rgba8" else torch.float64
    gath = [RowGather(height, width, 4, gdtype, gdev, rank, world, dist, always_collective=multi) for _ in range(n_buf)]
    rows_max = gath[0].rows_max
    direct = args.gather == "f64" and not rehearsal          # render straight into the gather buffer
    frames64 = [g.local for g in gath] if direct else [torch.zeros((rows_max, width, 4), dtype=torch.float64, device="cuda")]
    stage8 = torch.zeros((rows_max, width, 4), dtype=torch.uint8, device="cuda") if (rehearsal and args.gather == "rgba8") else None
    state = {"k": 0, "pending": None, "pending_buf": 0}
    torch.cuda.synchronize()                # the buffers above are filled on torch's stream, the renderer has its own

    # GPU runs: an event per gather buffer marks "the gather that read this buffer has finished"; the host waits for it
    # only when the buffer comes round again, two frames later, so rank 0's de-interleave (N strided copies) runs beside
    # the next frame instead of in front of it.
    ev_free = [torch.cuda.Event() for _ in range(n_buf)] if (multi and not rehearsal) else None
    # the renderer's stream as torch sees it: the gather is ordered behind the quantisation without stopping the host
    ctx_stream = torch.cuda.ExternalStream(gpu.stream) if (multi and not rehearsal) else None
    ev_set = [False] * n_buf

    def drain():
        g = state["pending"]
        if g is None:
            return
        if rehearsal:
            g.finish()
        else:
            b = state["pending_buf"]
            g.wait()                        # torch's stream now runs behind the gather ...
            ev_free[b].record()             # ... this marks its end ...
            ev_set[b] = True
            g.deinterleave()                # ... and the de-interleave follows it on that stream: the host waits for neither
        state["pending"] = None

    def step(profile):
        k = state["k"]
        state["k"] = k + 1
        b = k % n_buf
        g = gath[b]
        f64 = frames64[k % len(frames64)]
        if direct and ev_free is not None and ev_set[b]:
            ev_free[b].synchronize()        # --gather f64 renders straight into the gather buffer: it must be free already
        if args.selftest_gather and args.verify:
            f64.fill_(-1.0)                 # every frame is the same image: poison what it is written into, so that a
            torch.cuda.synchronize()        # buffer read too early or too late shows in --verify
        st = gpu.render_device(f64.data_ptr(), width, height, depth, row_begin=rank, row_step=world, profile=profile)
        if multi:
            drain()                         # frame k-1: gathered while frame k was rendered
            if ev_free is not None and ev_set[b]:
                ev_free[b].synchronize()    # the gather of frame k-2 read this buffer; it ended a frame ago
            if args.selftest_gather and args.verify and args.gather == "rgba8":
                g.local.fill_(171)
                torch.cuda.synchronize()
            if args.gather == "rgba8":
                dst8 = stage8 if rehearsal else g.local
                gpu.quantize_device(f64.data_ptr(), dst8.data_ptr(), rows_max * width)
                if ctx_stream is not None:
                    torch.cuda.current_stream().wait_stream(ctx_stream)
                else:
                    gpu.synchronize()
                if rehearsal:
                    g.local.copy_(stage8)
            elif rehearsal:
                g.local.copy_(f64)
            g.start()
            state["pending"] = g
            state["pending_buf"] = b
        return st

    def fence():
        drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(1 if w == args.warmup - 1 else 0)      # (the last one with the launch events: their one-time creation is not a step's cost)
    fence()
    t0 = time.perf_counter()
    agg = {"traced": 0, "ref_equiv": 0, "trace_ms": 0.0, "launches": 0, "primary": 0, "secondary": 0, "shadow": 0,
           "frame_ms": 0.0, "levels": 0, "profiled_steps": 0, "profiled_launches": 0, "profiled_rays": 0}
    every = max(1, args.profile_every)
    for i in range(args.steps):
        profiled = (i % every) == 0
        st = step(1 if profiled else 0)
        rays = st.rays_primary + st.rays_secondary + st.rays_shadow
        agg["traced"] += rays
        agg["primary"] += st.rays_primary
        agg["secondary"] += st.rays_secondary
        agg["shadow"] += st.rays_shadow
        agg["ref_equiv"] += st.rays_ref_equiv
        agg["launches"] += st.trace_launches
        agg["levels"] = max(agg["levels"], st.levels)
        if profiled:
            # the steps whose trace launches carried HIP events (on the renderer's stream, inside the timed region)
            agg["profiled_steps"] += 1
            agg["profiled_launches"] += st.trace_launches
            agg["profiled_rays"] += rays
            agg["trace_ms"] += st.trace_ms
            agg["frame_ms"] += st.frame_ms
    fence()
    elapsed = time.perf_counter() - t0

    counts = torch.tensor([agg["traced"], agg["ref_equiv"]], dtype=torch.float64, device=gdev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
    if multi:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_traced, total_ref = counts.tolist()
    elapsed = float(tmax.item())

    if rank == 0:
        steps = max(1, args.steps)
        bytes_per_ray = algorithmic_bytes_per_ray(dims)
        avg_launch_ms = agg["trace_ms"] / max(1, agg["profiled_launches"])
        rays_per_launch = agg["profiled_rays"] / max(1, agg["profiled_launches"])
        achieved = (rays_per_launch * bytes_per_ray) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        ms_per_step = elapsed / steps * 1e3
        launches_per_step = agg["launches"] / steps
        stream = launches_per_step <= 1.0 + 1e-9        # the streaming frame kernel: one launch does the whole ray tree
        prof, fresh = committed_profile(args.workload, width, height) if world == 1 else (None, None)
        roofline = {
            # the contract's figure: ALGORITHMIC bytes (64N + 168 per ray, the ray and hit records written and
            # read once, the framebuffer read-modify-write) of the rays one launch of the dominant kernel traces, over
            # that kernel's average duration.  The bytes of B_ray are moved by ALL kernels of the frame, the time is one
            # kernel's: `frac_frame` divides by the whole frame instead, and `measured` is what the counters saw.
            # `bound` is the roofline the contract prices the path against (BASELINE.json: HBM, 8 TB/s); `bound_that_holds` is
            # what the counters say limits the kernel -- it is NOT HBM-bound (measured traffic: single-digit % of peak, see
            # `measured`): FP64 vector issue and the latency of a divergent traversal (`secondary`)
            "bound": "hbm",
            "bound_that_holds": "fp64_valu_issue_and_latency",
            "kernel": "k_frame_stream (the whole ray tree in one persistent launch)" if stream else "k_trace (trace_kd, one ray per lane)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_frame": (total_traced / steps / max(1, world)) * bytes_per_ray / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": prof["hbm_bytes_per_launch"] if (prof and fresh and "hbm_bytes_per_launch" in prof) else None,
            "traffic_unit": "HBM bytes per launch of that kernel: rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes of this "
                            "command (profiles/profile_workload.sh), quoted only when measured on this very libndt_hip.so",
            "profile_stale": (None if prof is None else (not fresh)),
            "algorithmic_bytes_per_launch": rays_per_launch * bytes_per_ray,
            "bytes_per_ray": bytes_per_ray,
            "kernel_own_bytes_per_ray": 16 * dims + 16,     # what the trace kernel itself must move: o, v, dist_limit in; object, primitive out
            "rays_per_launch": rays_per_launch,
            "avg_launch_ms": avg_launch_ms,
            "launches_per_step": launches_per_step,
            "profiled_steps": agg["profiled_steps"],
            "timing": "HIP events on the renderer's stream carrying the kernels' own dispatch timestamps, on every "
                      "%s timed step" % ("" if every == 1 else "%d-th" % every),
        }
        if prof and fresh:
            if "hbm_bytes_per_launch" in prof:
                gbs = prof["hbm_bytes_per_launch"] / (prof["avg_launch_ns"] * 1e-9) / 1e9
                roofline["measured"] = {"hbm_gbs": gbs, "frac": gbs / HBM_PEAK_GBS,
                                        "bytes_per_ray": prof["hbm_bytes_per_launch"] / max(1.0, rays_per_launch),
                                        "avg_launch_ms_under_rocprof": prof["avg_launch_ns"] * 1e-6}
            if "fp64_valu_issue_frac" in prof:
                # the bound that actually holds: FP64 vector issue and the latency of a divergent traversal
                roofline["secondary"] = {"bound": "fp64_valu_issue", "frac": prof["fp64_valu_issue_frac"],
                                         "lane_util": prof.get("lane_utilisation"), "salu_per_valu": prof.get("salu_per_valu"),
                                         "wait_frac_of_wave_cycles": prof.get("wait_frac_of_wave_cycles"),
                                         "source": "profiles/r04_profile_%s_%dp.json" % (args.workload, height)}
        line = {
            "metric": "Mray/s (primary+shadow+reflect) at 1920x1080",
            "value": total_traced / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[%d]: %s scene (tests/golden/%s.ndtscene.gz), %d-D, %dx%d, -l %d, "
                            "samples=1, mono, kd-tree on, specular on; rows cyclic over %d GPU(s)%s" % (
                                cfg_idx, args.workload, fixture, dims, width, height, depth, world,
                                ", %s gather of the %s image to rank 0" % ("RCCL" if not rehearsal else "gloo (rehearsal)", args.gather)
                                if multi else ""),
                "width": width, "height": height, "dims": dims, "max_optic_depth": depth,
                "parallelism": "rows%d" % world,
            },
            "rays_traced_per_step": total_traced / steps,
            "rays_ref_equiv_per_step": total_ref / steps,
            "mray_s_ref_equiv": total_ref / elapsed / 1e6,
            "ray_mix_rank0_per_step": {"primary": agg["primary"] / steps, "secondary": agg["secondary"] / steps,
                                        "shadow": agg["shadow"] / steps, "bounces": agg["levels"]},
            "device_frame_ms_rank0": agg["frame_ms"] / max(1, agg["profiled_steps"]),
            "roofline": roofline,
        }
        if world == 1:
            # what a caller of ndt_hip_render_rgba8 waits for: render + pixel_d2c on the device + the 4-bytes-per-pixel
            # copy to host memory (the reference's "rendering took" ends with the pixels in host memory)
            gpu.render_rgba8(width, height, depth)
            t1 = time.perf_counter()
            for _ in range(5):
                gpu.render_rgba8(width, height, depth)
            line["ms_per_step_host_rgba8"] = (time.perf_counter() - t1) / 5 * 1e3
            # ... and in a SEQUENCE of frames (an animation, the reference's real use): ndt_hip_render_rgba8_async -- the copy of
            # frame k to pinned host memory runs behind the rendering of frame k + 1.  20 frames, the last one's arrival included.
            pinned = [torch.empty((height, width, 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
            gpu.render_rgba8_async(pinned[0].data_ptr(), width, height, depth)
            gpu.render_rgba8_wait()
            n_seq = 20
            t1 = time.perf_counter()
            for k in range(n_seq):
                gpu.render_rgba8_async(pinned[k & 1].data_ptr(), width, height, depth)
            gpu.render_rgba8_wait()
            line["ms_per_step_host_rgba8_pipelined"] = (time.perf_counter() - t1) / n_seq * 1e3
            check, _ = gpu.render_rgba8(width, height, depth)
            line["host_rgba8_pipelined_bytes_identical"] = bool(np.array_equal(pinned[(n_seq - 1) & 1].numpy(), check))
            # ... and with TWO frames in flight (two contexts, two host threads -- the host program's -j 2, the reference's
            # frame-level parallelism): the launches of one frame fill the tails of the other's.  Not the headline: a frame alone
            # takes ms_per_step.  2 x 10 frames, images left in HBM like the timed steps.
            import threading
            second = NdtHip(local_rank)
            second.upload_scene(fs)
            pair = [(gpu, f64_frames_in_flight(torch, height, width)), (second, f64_frames_in_flight(torch, height, width))]

            def frames(ctx, buf, n):
                for _ in range(n):
                    ctx.render_device(buf.data_ptr(), width, height, depth)

            for ctx, buf in pair:
                frames(ctx, buf, 2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ths = [threading.Thread(target=frames, args=(ctx, buf, 10)) for ctx, buf in pair]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize()
            line["ms_per_step_two_frames_in_flight"] = (time.perf_counter() - t1) / 20 * 1e3
            line["two_frames_in_flight_identical"] = bool(torch.equal(pair[0][1], pair[1][1]))
            second.close()
        elif args.scaling == "strong" and not rehearsal:
            # the time this frame takes on ONE GPU (rank 0's, alone: the other ranks wait at the barrier below)
            whole = torch.zeros((height, width, 4), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            gpu.render_device(whole.data_ptr(), width, height, depth)
            t1 = time.perf_counter()
            for _ in range(5):
                gpu.render_device(whole.data_ptr(), width, height, depth)
            gpu.synchronize()
            one = (time.perf_counter() - t1) / 5 * 1e3
            line["strong_reference"] = {"ms_one_gpu_same_frame": one, "speedup": one / line["ms_per_step"],
                                        "what": "the whole %dx%d frame rendered on rank 0's GPU alone (image left in HBM), "
                                                "5 frames after the timed region" % (width, height)}
            del whole
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, width, height, depth, args.cpu_threads)
            line["cpu_baseline_t1"] = cpu_baseline(args.workload, width, height, depth, 1)
        print(json.dumps(line), flush=True)
    if args.verify and multi and rank == 0:
        whole = torch.zeros((height, width, 4), dtype=torch.float64, device="cuda")
        want = torch.zeros((height, width, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()            # torch fills on its stream, the renderer writes on its own
        gpu.render_device(whole.data_ptr(), width, height, depth)
        last = gath[(state["k"] - 1) % n_buf].image
        if args.gather == "rgba8":
            gpu.quantize_device(whole.data_ptr(), want.data_ptr(), height * width)
            gpu.synchronize()
        else:
            want = whole
        same = bool(torch.equal(last.cpu(), want.cpu()))
        if not same:
            bad = (last.cpu() != want.cpu()).any(dim=2)
            rows = bad.any(dim=1).nonzero().flatten()
            print("verify: %d pixels differ, rows %s ..." % (int(bad.sum()), rows[:8].tolist()), file=sys.stderr)
        print("verify: gathered frame %s the single-GPU render" % ("==" if same else "!="), file=sys.stderr, flush=True)
        if not same:
            raise SystemExit("bench.py --verify: the gathered frame differs")
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    gpu.close()


if __name__ == "__main__":
    main()
