"""One frame split over K contexts ON ONE GPU (ndt_hip_render_multi: rows dealt cyclically, each context its own stream and
workspace): the GPU runs the contexts' kernels side by side, so the tail of one context's trace launch overlaps the bulk of
another's.  usage: [NDT_HIP_PIPELINE=levels] python profiles/multi_ctx_probe.py [workload] [WxH] [K ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip, render_multi, IMAGE_F64, IMAGE_RGBA8

FIX = {"random4d": ("c3_random4d", 4), "balls4d": ("c2_balls4d", 128), "hypercube3d": ("c1_hypercube3d", 128),
       "hypercube6d": ("c5_hypercube6d", 128), "hypercube8d": ("c5_hypercube8d", 128)}
name = sys.argv[1] if len(sys.argv) > 1 else "random4d"
w, h = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x")]
ks = [int(x) for x in sys.argv[3:]] or [1, 2, 3, 4, 6]
fix, depth = FIX[name]
fs = load_scene("tests/golden/%s.ndtscene.gz" % fix)
ctxs = [NdtHip(0) for _ in range(max(ks))]
for c in ctxs:
    c.upload_scene(fs)
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
ref = None
for k in ks:
    for _ in range(3):
        render_multi(ctxs[:k], w, h, depth, IMAGE_F64, d_out_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        _, st = render_multi(ctxs[:k], w, h, depth, IMAGE_F64, d_out_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    img = buf.cpu()
    if ref is None:
        ref = img
    print("%s %dx%d pipeline=%s K=%d: %.3f ms per frame, same image: %s" % (
        name, w, h, os.environ.get("NDT_HIP_PIPELINE", "stream"), k, ms, bool(torch.equal(img, ref))), flush=True)
