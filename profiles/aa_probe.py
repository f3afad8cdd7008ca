"""Recursive anti-aliasing on the device: time and statistics at full resolution (development aid).
usage: python profiles/aa_probe.py [scene] [WxH] [diff,depth] [-l depth]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

scene = sys.argv[1] if len(sys.argv) > 1 else "c3_random4d"
w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
aa = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "20,4").split(","))
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
fs = load_scene("tests/golden/%s.ndtscene.gz" % scene)
g = NdtHip(0)
g.upload_scene(fs)
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
for label, kw in (("plain", {}), ("aa %d,%d" % aa, {"aa": aa})):
    g.render_device(buf.data_ptr(), w, h, depth, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        st = g.render_device(buf.data_ptr(), w, h, depth, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    d = st.as_dict()
    print("%s %dx%d %s: %.2f ms per frame; traced %d rays (reference-equivalent %d), %d pixels resampled, %d extra samples" % (
        scene, w, h, label, ms, d["rays_primary"] + d["rays_secondary"] + d["rays_shadow"], d["rays_ref_equiv"],
        d["pixels_resampled"], d["aa_samples"]))
