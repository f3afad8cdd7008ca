"""-n samples > 1 on the device vs the oracle (which follows the reference's drand48 stream): how the
difference shrinks with the number of samples, and what a frame costs (development aid)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
from conftest import Oracle, golden
from ndt_amd.hip import NdtHip

g = golden(sys.argv[1] if len(sys.argv) > 1 else "ns_zoo4d_dof")
o = Oracle()
gpu = NdtHip(0)
gpu.upload_scene(g.scene)
for S in (2, 8, 32, 128):
    want, so = o.render(g.scene, g.width, g.height, g.depth, samples=S, seed48=g.meta["seed48"])
    out, st = gpu.render(g.width, g.height, g.depth, samples=S)
    print("%s %dx%d -n %d: mean |device - oracle| %.5f, image means %.5f / %.5f, samples per pixel device %.1f oracle %.1f" % (
        g.name, g.width, g.height, S, np.abs(out[..., :3] - want[..., :3]).mean(), out[..., :3].mean(), want[..., :3].mean(),
        st.aa_samples / (g.width * g.height), so.rays_primary / (g.width * g.height)))
w, h = 1920, 1080
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
for S in (4, 16):
    gpu.render_device(buf.data_ptr(), w, h, g.depth, samples=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = gpu.render_device(buf.data_ptr(), w, h, g.depth, samples=S)
    torch.cuda.synchronize()
    d = st.as_dict()
    print("%s 1920x1080 -n %d: %.1f ms, %d rays traced, %.1f samples per pixel" % (
        g.name, S, 1e3 * (time.perf_counter() - t0), d["rays_primary"] + d["rays_secondary"] + d["rays_shadow"], d["aa_samples"] / (w * h)))
