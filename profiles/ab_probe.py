"""The same frames on whatever library NDT_HIP_LIB names (default: ndt_amd/libndt_hip.so): run it once per library on the same
box to compare builds.  Default pipeline choice (auto), default options.
usage: [NDT_HIP_LIB=...] python profiles/ab_probe.py [--quick]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip, LIB_PATH

cases = [("c3_random4d", 4, 1920, 1080, 1), ("c2_balls4d", 4, 1920, 1080, 1), ("c1_hypercube3d", 128, 1920, 1080, 1),
         ("c3_random4d", 4, 3840, 2160, 8), ("c3_random4d", 4, 3840, 2160, 1), ("c3_random4d", 4, 960, 540, 1), ("c3_random4d", 4, 64, 36, 1),
         ("c5_hypercube6d", 128, 1920, 1080, 1), ("c5_hypercube7d", 128, 1920, 1080, 1), ("c5_hypercube8d", 128, 1920, 1080, 1)]
if "--quick" in sys.argv:
    cases = cases[:4]
buf = torch.empty((2160, 3840, 4), dtype=torch.float64, device="cuda")
print("library: %s" % os.path.basename(LIB_PATH))
last = None
g = None
for scene, depth, w, h, shard in cases:
    if scene != last:
        if g:
            g.close()
        g = NdtHip(0)
        g.upload_scene(load_scene("tests/golden/%s.ndtscene.gz" % scene))
        last = scene
    for _ in range(3):
        g.render_device(buf.data_ptr(), w, h, depth, row_begin=0, row_step=shard)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            st = g.render_device(buf.data_ptr(), w, h, depth, row_begin=0, row_step=shard)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / n)
    print("%-16s %4dx%-4d r::%d: %.3f ms a frame (%d trace launches)" % (scene, w, h, shard, best, st.trace_launches), flush=True)
