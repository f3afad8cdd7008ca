"""The zoo's frame time (480x270 -l 5) against the culls of its hcube's faces: the index by thin hull axes (face_groups), the
hierarchy over the face boxes (face_tree), or the boxes 63 at a time.  usage: python profiles/zoo_cull_probe.py [scene ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
w, h = 480, 270
names = sys.argv[1:] or ["zoo5d_f2", "zoo6d", "zoo9d", "zoo10d"]
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
for name in names:
    fs = load_scene("tests/golden/%s.ndtscene.gz" % name)
    line = "%-9s" % name
    ref = None
    for groups, tree in ((1, 1), (0, 1), (0, 0)):
        if name == "zoo10d" and not tree:
            continue
        g = NdtHip(0)
        g.set_option("face_groups", groups)
        g.set_option("face_tree", tree)
        g.upload_scene(fs)
        g.render_device(buf.data_ptr(), w, h, 5)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            g.render_device(buf.data_ptr(), w, h, 5)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        if ref is None:
            ref = buf.clone()
        same = bool(torch.equal(ref, buf))
        line += "  groups %d tree %d: %7.2f ms%s" % (groups, tree, best, "" if same else " (DIFFERENT IMAGE)")
        g.close()
    print(line, flush=True)
