"""Do two renders of the same frame on one GPU overlap?  (development aid)
usage: python profiles/concurrent_probe.py [n_contexts] [frames]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fs = load_scene("tests/golden/c3_random4d.ndtscene.gz")
w, h = 1920, 1080
ctxs = [NdtHip(0) for _ in range(n_ctx)]
bufs = [torch.empty((h // 1, w, 4), dtype=torch.float64, device="cuda") for _ in range(n_ctx)]
for c in ctxs:
    c.upload_scene(fs)


def work(i, step, reps):
    rows = (h - i + step - 1) // step if step > 1 else h
    for _ in range(reps):
        ctxs[i].render_device(bufs[i].data_ptr(), w, h, 4, row_begin=i if step > 1 else 0, row_step=step)


for mode in ("whole frames", "row shards of one frame"):
    step = 1 if mode == "whole frames" else n_ctx
    for i in range(n_ctx):
        work(i, step, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i, step, frames)) for i in range(n_ctx)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total_frames = frames * (n_ctx if step == 1 else 1)
    print("%d contexts, %s: %.3f ms per full frame" % (n_ctx, mode, 1e3 * dt / total_frames))
