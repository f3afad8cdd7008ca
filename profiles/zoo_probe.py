"""Frame time of the zoo in 3-D .. 10-D (its hcube has 2 .. 52 904 nested faces; more than 63: no face boxes).  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
w, h = 480, 270
for name in ("zoo3d_mirror", "zoo4d", "zoo5d_f2", "zoo6d", "zoo9d", "zoo10d", "zoo11d"):
    fs = load_scene("tests/golden/%s.ndtscene.gz" % name)
    g = NdtHip(0); g.upload_scene(fs)
    buf = torch.empty((h, w, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    g.render_device(buf.data_ptr(), w, h, 5)
    t0 = time.perf_counter()
    st = g.render_device(buf.data_ptr(), w, h, 5)
    ms = (time.perf_counter() - t0) * 1e3
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    nf = max([o["n_obj"] for o in fs.objects if o["type"] == 6] + [0])
    print("%-13s %d-D %dx%d -l 5: %9.2f ms, %8d rays, %8.1f Mray/s; %d objects, hcube faces %d" % (name, fs.dims, w, h, ms, rays, rays / ms / 1e3, fs.struct.n_objects, nf), flush=True)
    g.close()
