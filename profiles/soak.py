"""Soak: the same frames over and over, every image compared with the first of its kind (the frame kernel's queues and the
per-bounce pipeline must give the same bits every time, and never hang).  usage: python profiles/soak.py [seconds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cases = [("c3_random4d", 64, 36, 4, {}), ("c3_random4d", 480, 270, 4, {}), ("c3_random4d", 960, 540, 4, {}),
         ("c3_random4d", 1920, 1080, 4, {}), ("c2_balls4d", 640, 360, 128, {}), ("zoo3d_mirror", 320, 240, 128, {}),
         ("c5_hypercube6d", 480, 270, 128, {}), ("c3_random4d", 240, 135, 4, {"aa": (20, 3)}),
         ("st_zoo4d_sbs", 160, 90, 6, {"stereo": 1}), ("c5_hypercube8d", 480, 270, 128, {}), ("c5_hypercube7d", 320, 180, 128, {}),
         ("ns_zoo4d_dof", 96, 54, 6, {"samples": 4}), ("al_zoo4d", 96, 54, 5, {"samples": 2}), ("zoo10d", 120, 72, 5, {}),
         ("c3_random4d", 3840, 2160, 4, {})]
ctxs = {}
first = {}
count = {}
t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    for k, (scene, w, h, depth, kw) in enumerate(cases):
        if scene not in ctxs:
            g = NdtHip(0)
            g.upload_scene(load_scene("tests/golden/%s.ndtscene.gz" % scene))
            ctxs[scene] = g
        g = ctxs[scene]
        rows = h
        buf = torch.full((2 * h + 64, 2 * w, 4), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.current_stream().synchronize()       # torch fills on its stream, the renderer writes on its own
        st = g.render_device(buf.data_ptr(), w, h, depth, **kw)
        torch.cuda.synchronize()
        img = buf[:rows, :w].clone() if not kw.get("stereo") else buf.clone()
        rays = st.rays_primary + st.rays_secondary + st.rays_shadow
        if k not in first:
            first[k] = (img, rays)
            count[k] = 1
        else:
            assert torch.equal(img, first[k][0]) and rays == first[k][1], "case %d (%s %dx%d) differs in round %d" % (k, scene, w, h, rounds)
            count[k] += 1
    rounds += 1
    if rounds % 50 == 0:
        print("round %d, %.0f s left" % (rounds, t_end - time.time()), flush=True)
print("soak: %d rounds of %d cases, every frame identical to the first of its kind: %s" % (rounds, len(cases), [count[k] for k in sorted(count)]))
