"""-n samples at full size: one frame under the profiler (development aid).  usage: python profiles/ns_probe.py [S] [scene]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scene = sys.argv[2] if len(sys.argv) > 2 else "c3_random4d"
fs = load_scene("tests/golden/%s.ndtscene.gz" % scene)
g = NdtHip(0)
g.upload_scene(fs)
w, h = 1920, 1080
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
g.render_device(buf.data_ptr(), w, h, 4, samples=S)
torch.cuda.synchronize()
t0 = time.perf_counter()
st = g.render_device(buf.data_ptr(), w, h, 4, samples=S)
torch.cuda.synchronize()
d = st.as_dict()
print("%s 1920x1080 -n %d: %.2f ms, %d rays, %.1f samples a pixel" % (scene, S, 1e3 * (time.perf_counter() - t0),
      d["rays_primary"] + d["rays_secondary"] + d["rays_shadow"], d["aa_samples"] / (w * h)))
