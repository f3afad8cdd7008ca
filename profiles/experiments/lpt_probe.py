"""Experiment (make -C ndt_amd/csrc lpt; NDT_HIP_LIB=ndt_amd/libndt_hip_lpt.so): what would a PERFECT longest-batch-first order of every
trace launch's batches buy?  The same frame is rendered over and over: NDT_LPT=1 records every batch's duration and orders the next
frame's batches by it, NDT_LPT=2 applies the last order without recording.
usage: NDT_HIP_LIB=... python profiles/lpt_probe.py [scene] [WxH] [shard]"""
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
g = NdtHip(0)
g.upload_scene(load_scene("tests/golden/%s.ndtscene.gz" % scene))
g.set_option("pipeline", 1)
rows = (h + shard - 1) // shard
buf = torch.empty((rows, w, 4), dtype=torch.float64, device="cuda")


def timed(label):
    for _ in range(3):
        g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / 20)
    print("%s %dx%d r::%d %s: %.3f ms a frame" % (scene, w, h, shard, label, best), flush=True)


os.environ["NDT_LPT"] = "0"
timed("queue order")
ref = buf.clone()
g.set_option("debug_levels", 1)
g.set_option("exit_probe", 1)
for it in range(3):
    os.environ["NDT_LPT"] = "1"
    g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard, profile=1 if it == 2 else 0)
os.environ["NDT_LPT"] = "2"
g.render_device(buf.data_ptr(), w, h, 4, row_begin=0, row_step=shard, profile=1)
g.set_option("debug_levels", 0)
g.set_option("exit_probe", 0)
timed("longest batch first (perfect knowledge)")
print("image identical: %s" % bool(torch.equal(ref, buf)))
os.environ["NDT_LPT"] = "0"
timed("queue order again")
