"""Every BASELINE configuration at its stated size on the device, beside the time the compiled reference took for the same frame when
the fixture was generated (tests/golden/<case>.json: ref_render_s; its own pthreads on the container that generated the fixtures --
a different machine from the GPU box, so a ratio of two machines, not a same-box figure: bench.py's cpu_baseline is that).
usage: python profiles/baseline_configs_probe.py"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

cases = [("c1_hypercube3d_256", "c1_hypercube3d", "configs[0]"), ("c2_balls4d_1080p", "c2_balls4d", "configs[1]"),
         ("c3_random4d_1080p", "c3_random4d", "configs[2] (bench.py)"), ("c4_random4d_4k", "c3_random4d", "configs[3]"),
         ("c5_hypercube6d_1080p", "c5_hypercube6d", "configs[4] 6-D"), ("c5_hypercube7d_1080p", "c5_hypercube7d", "configs[4] 7-D"),
         ("c5_hypercube8d_1080p", "c5_hypercube8d", "configs[4] 8-D")]
buf = torch.empty((2160, 3840, 4), dtype=torch.float64, device="cuda")
for case, scene, label in cases:
    m = json.load(open("tests/golden/%s.json" % case))
    w, h, depth = m["width"], m["height"], m["depth"]
    g = NdtHip(0)
    g.upload_scene(load_scene("tests/golden/%s.ndtscene.gz" % scene))
    for _ in range(3):
        st = g.render_device(buf.data_ptr(), w, h, depth)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            st = g.render_device(buf.data_ptr(), w, h, depth)
        torch.cuda.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t0) / 10)
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    print("%-22s %-20s %d-D %4dx%-4d -l %-3d: %7.3f ms, %5.2f M rays traced (%6.1f M in the reference's counting = the fixture's %6.1f M); "
          "the reference took %6.2f s for it" % (label, scene, m["dims"], w, h, depth, best, rays / 1e6, st.rays_ref_equiv / 1e6,
                                                m["rays_total"] / 1e6, m["ref_render_s"]), flush=True)
    g.close()
