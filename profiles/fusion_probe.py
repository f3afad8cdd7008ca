"""Frame time with and without the first trace launch making its own primaries (option fuse_primaries), per-bounce kernels.
usage: python profiles/fusion_probe.py [WxH] [shard] [--opt=name]   (default option: fuse_primaries; --auto: the default pipeline choice; --set=name=value: other options)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = ([a[6:] for a in sys.argv[1:] if a.startswith("--opt=")] or ["fuse_primaries"])[0]
w, h = (int(x) for x in (args[0] if args else "1920x1080").split("x"))
shard = int(args[1]) if len(args) > 1 else 1
rows = (h + shard - 1) // shard
buf = torch.empty((rows, w, 4), dtype=torch.float64, device="cuda")
for scene, depth in (("c3_random4d", 4), ("c2_balls4d", 4), ("c1_hypercube3d", 128), ("c5_hypercube6d", 128), ("c5_hypercube8d", 128)):
    fs = load_scene("tests/golden/%s.ndtscene.gz" % scene)
    g = NdtHip(0)
    g.upload_scene(fs)
    if "--auto" not in sys.argv:
        g.set_option("pipeline", 1)
    for kv in [a[6:] for a in sys.argv[1:] if a.startswith("--set=")]:
        g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for fuse in (0, 1, 0, 1):
        g.set_option(opt, fuse)
        for _ in range(3):
            g.render_device(buf.data_ptr(), w, h, depth, row_begin=0, row_step=shard)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                st = g.render_device(buf.data_ptr(), w, h, depth, row_begin=0, row_step=shard)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / n)
        print("%s %dx%d r::%d %s %d: %.3f ms a frame" % (scene, w, h, shard, opt, fuse, best), flush=True)
    g.close()
