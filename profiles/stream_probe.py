"""What every wavefront of the streaming frame kernel did (NDT_HIP_STREAM_PROBE): items per kind, time per item, idle
rounds, when the last item finished.  usage: python profiles/stream_probe.py [workload] [WxH ...]"""
import os
import sys
os.environ["NDT_HIP_STREAM_PROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

FIX = {"random4d": ("c3_random4d", 4), "balls4d": ("c2_balls4d", 128), "hypercube3d": ("c1_hypercube3d", 128),
       "hypercube6d": ("c5_hypercube6d", 128), "hypercube8d": ("c5_hypercube8d", 128)}
name = sys.argv[1] if len(sys.argv) > 1 else "random4d"
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[2:]] or [(1920, 1080)]
fix, depth = FIX[name]
fs = load_scene("tests/golden/%s.ndtscene.gz" % fix)
g = NdtHip(0)
g.upload_scene(fs)
for w, h in sizes:
    buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
    g.render_device(buf.data_ptr(), w, h, depth)
    g.render_device(buf.data_ptr(), w, h, depth)
    print("== %s %dx%d" % (name, w, h), file=sys.stderr, flush=True)
    st = g.render_device(buf.data_ptr(), w, h, depth, profile=1)
    print("   frame %.3f ms, kernel %.3f ms" % (st.frame_ms, st.trace_ms), file=sys.stderr, flush=True)
