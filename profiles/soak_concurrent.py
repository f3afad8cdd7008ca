"""Soak with K host threads, each with its own contexts, rendering at the same time (the host program's -j K): persistent kernels
of different contexts share the GPU; every image must equal the one a lone context makes.  usage: python profiles/soak_concurrent.py [K] [seconds]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
cases = [("c3_random4d", 64, 36, 4, {}), ("c3_random4d", 960, 540, 4, {}), ("c3_random4d", 1920, 1080, 4, {}),
         ("c2_balls4d", 640, 360, 128, {}), ("c5_hypercube6d", 480, 270, 128, {}), ("c3_random4d", 240, 135, 4, {"aa": (20, 3)})]
scenes = {s: load_scene("tests/golden/%s.ndtscene.gz" % s) for s in {c[0] for c in cases}}


def render(ctxs, case):
    scene, w, h, depth, kw = case
    if scene not in ctxs:
        ctxs[scene] = NdtHip(0)
        ctxs[scene].upload_scene(scenes[scene])
    buf = torch.full((h, w, 4), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.current_stream().synchronize()       # torch fills on its stream, the renderer writes on its own
    st = ctxs[scene].render_device(buf.data_ptr(), w, h, depth, **kw)
    ctxs[scene].synchronize()
    return buf, st.rays_primary + st.rays_secondary + st.rays_shadow


lone = {}
want = [render(lone, c) for c in cases]
torch.cuda.synchronize()
t_end = time.time() + budget
done = [0] * K
bad = []


def work(k):
    ctxs = {}
    i = k
    while time.time() < t_end and not bad:
        c = i % len(cases)
        img, rays = render(ctxs, cases[c])
        if rays != want[c][1] or not torch.equal(img, want[c][0]):
            bad.append((k, c))
        done[k] += 1
        i += 1 + k          # every thread walks the cases with its own stride


ths = [threading.Thread(target=work, args=(k,)) for k in range(K)]
for t in ths:
    t.start()
for t in ths:
    t.join()
assert not bad, "thread %d, case %d: image differs from the lone context's" % bad[0]
print("concurrent soak: %d threads, %s frames each in %.0f s, every image identical to a lone context's" % (K, done, budget))
