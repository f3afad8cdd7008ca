"""Device against oracle: image and reference-equivalent ray count for mono / side-by-side / over-under, with and without
recursive anti-aliasing (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import Oracle, golden
from ndt_amd.hip import NdtHip
g = golden("st_zoo4d_sbs")
o = Oracle()
gpu = NdtHip(0)
gpu.upload_scene(g.scene)
for res in ((64, 36), (48, 54), (40, 30)):
    for stereo in (0, 1, 2):
        for aa in (None, (8, 2), (8, 3)):
            out, st = gpu.render(res[0], res[1], 6, aa=aa, stereo=stereo)
            want, so = o.render(g.scene, res[0], res[1], 6, aa=aa, stereo=stereo)
            print(res, "stereo", stereo, "aa", aa, "max diff %.2e" % np.abs(out - want).max(), "rays", st.rays_ref_equiv, so.rays_ref_equiv,
                  st.rays_ref_equiv - so.rays_ref_equiv, "unique", st.rays_primary + st.rays_secondary + st.rays_shadow,
                  so.rays_primary + so.rays_secondary + so.rays_shadow, flush=True)
