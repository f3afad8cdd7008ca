"""What ONE rank of an N-GPU run renders, timed on one GPU: rows rank::N of BASELINE configs[3]'s 3840x2160 frame (strong scaling:
bench.py --gpus N renders exactly this per rank) -- the predicted N-GPU frame time is the slowest shard (+ the 8-bit gather,
4 bytes per pixel over xGMI: under 0.1 ms).  --weak: the N x 1080p frames of `bench.py --scaling weak` instead.
usage: python profiles/shard_probe.py [--weak] [N ...]"""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

args = [a for a in sys.argv[1:] if not a.startswith("--")]
weak = "--weak" in sys.argv
fs = load_scene("tests/golden/c3_random4d.ndtscene.gz")
g = NdtHip(0)
g.upload_scene(fs)


def timed(w, h, rank, n):
    rows = (h - rank + n - 1) // n
    buf = torch.empty((rows, w, 4), dtype=torch.float64, device="cuda")
    for _ in range(3):
        g.render_device(buf.data_ptr(), w, h, 4, row_begin=rank, row_step=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        st = g.render_device(buf.data_ptr(), w, h, 4, row_begin=rank, row_step=n)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, st


one = None
for n in [int(x) for x in args] or [1, 2, 4, 8]:
    if weak:
        s = math.sqrt(n)
        w, h = int(round(1920 * s / 8.0)) * 8, int(round(1080 * s / 8.0)) * 8
    else:
        w, h = 3840, 2160
    worst = 0.0
    for rank in sorted({0, n // 2, n - 1}):
        ms, st = timed(w, h, rank, n)
        worst = max(worst, ms)
        rays = st.rays_primary + st.rays_secondary + st.rays_shadow
        print("N=%d frame %dx%d rank %d: %.3f ms, %d rays, %.0f Mray/s, %d trace launch(es)" % (n, w, h, rank, ms, rays, rays / ms / 1e3, st.trace_launches))
    if n == 1:
        one = worst
    elif one and not weak:
        print("   -> predicted %d-GPU frame %.3f ms (slowest shard) = %.2fx the one-GPU frame (%.3f ms); whole-job %.0f Mray/s" % (
            n, worst, one / worst, one, 19518844 / worst / 1e3))
