"""Frame time against frame size: what part of a frame is fixed latency (development aid)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

fs = load_scene("tests/golden/%s.ndtscene.gz" % (sys.argv[1] if len(sys.argv) > 1 else "c3_random4d"))
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = NdtHip(0)
g.upload_scene(fs)
buf = torch.empty((2160, 3840, 4), dtype=torch.float64, device="cuda")
for w, h in ((64, 36), (240, 135), (480, 270), (960, 540), (1152, 648), (1280, 720), (1408, 792), (1920, 1080), (2716, 1528), (3840, 2160)):
    for _ in range(3):
        g.render_device(buf.data_ptr(), w, h, depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        st = g.render_device(buf.data_ptr(), w, h, depth)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    print("%4dx%-4d: %.3f ms, %9d rays, %.0f Mray/s" % (w, h, ms, rays, rays / ms / 1e3))
