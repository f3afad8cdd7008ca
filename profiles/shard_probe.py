"""What one rank of an N-GPU bench.py run renders (rows rank::N of the N x 1080p frame), timed on one GPU
(development aid).  usage: python profiles/shard_probe.py [N ...]"""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

fs = load_scene("tests/golden/c3_random4d.ndtscene.gz")
g = NdtHip(0)
g.upload_scene(fs)
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    s = math.sqrt(n)
    w, h = int(round(1920 * s / 8.0)) * 8, int(round(1080 * s / 8.0)) * 8
    rows = (h + n - 1) // n
    buf = torch.empty((rows, w, 4), dtype=torch.float64, device="cuda")
    for rank in sorted({0, n - 1}):
        for _ in range(3):
            g.render_device(buf.data_ptr(), w, h, 4, row_begin=rank, row_step=n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            st = g.render_device(buf.data_ptr(), w, h, 4, row_begin=rank, row_step=n)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        rays = st.rays_primary + st.rays_secondary + st.rays_shadow
        print("N=%d frame %dx%d rank %d: %.3f ms, %d rays, %.0f Mray/s" % (n, w, h, rank, ms, rays, rays / ms / 1e3))
