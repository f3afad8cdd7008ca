"""Cooperative stragglers (ndt_device.hpp:coop_trace): frame time against the budget after which a batch of the trace kernel
gives its last rays up, how many it may give up, whether only in the tail of a launch, and how many consumers a workgroup keeps.
usage: python profiles/coop_probe.py [scene] [WxH] [shard] [--probe | --grid]   (--probe: one profiled frame per setting with the exit probe)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

args = [a for a in sys.argv[1:] if not a.startswith("--")]
scene = args[0] if args else "c3_random4d"
w, h = (int(x) for x in (args[1] if len(args) > 1 else "1920x1080").split("x"))
shard = int(args[2]) if len(args) > 2 else 1
fs = load_scene("tests/golden/%s.ndtscene.gz" % scene)
g = NdtHip(0)
g.upload_scene(fs)
g.set_option("pipeline", 1)
rows = (h + shard - 1) // shard
buf = torch.empty((rows, w, 4), dtype=torch.float64, device="cuda")
# (on, budget us, rays at most, tail only, consumer wavefronts per workgroup)
settings = [(0, 0, 0, 1, 4)]
if "--grid" in sys.argv:
    settings += [(1, b, k, 1, cw) for cw in (4, 12) for b in (5, 10, 15, 20, 30) for k in (8, 16, 32, 64)]
    settings += [(1, b, k, 0, 4) for b in (60, 80) for k in (2, 4)]
else:
    settings += [(1, 10, 16, 1, 4), (1, 10, 32, 1, 12), (1, 20, 64, 1, 12)]
if "--off-only" in sys.argv:
    settings = settings[:1]
if "--probe" in sys.argv:
    g.set_option("exit_probe", 1)
    g.set_option("debug_levels", 1)
for on, budget, live, tail_only, cw in settings:
    g.set_option("coop", on)
    g.set_option("coop_budget_us", budget)
    g.set_option("coop_max_live", live)
    g.set_option("coop_tail_only", tail_only)
    g.set_option("coop_waves", cw)
    for _ in range(3):
        g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            st = g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    print("%s %dx%d r::%d coop %d budget %2d us, at most %2d rays, tail only %d, %2d consumers a workgroup: %.3f ms a frame" % (
        scene, w, h, shard, on, budget, live, tail_only, cw, best), flush=True)
    if "--probe" in sys.argv:
        sys.stderr.flush()
        st = g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard, profile=1)
        print("   profiled: frame %.3f ms, trace %.3f ms in %d launches" % (st.frame_ms, st.trace_ms, st.trace_launches), flush=True)
