"""A few tiny frames for a kernel timeline of the fixed latencies (development aid)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
fs = load_scene("tests/golden/c3_random4d.ndtscene.gz")
g = NdtHip(0)
g.upload_scene(fs)
w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64x36").split("x"))
buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
for _ in range(6):
    g.render_device(buf.data_ptr(), w, h, 4)
torch.cuda.synchronize()
