"""Device ensembles for offline comparison with a ground truth (development aid): K renders with different sample_seed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from conftest import golden
from ndt_amd.hip import NdtHip
gpu = NdtHip(0)
K = 96
out = {}
for name in sys.argv[1:]:
    g = golden(name)
    gpu.upload_scene(g.scene)
    for S in (64,):
        imgs = []
        for k in range(K):
            gpu.set_option("sample_seed", k + 1000)
            img, st = gpu.render(g.width, g.height, g.depth, samples=S, stereo=g.meta.get("stereo", 0))
            imgs.append(img)
        imgs = np.array(imgs)
        out["%s_S%d_mean" % (name, S)] = imgs.mean(axis=0)
        out["%s_S%d_var" % (name, S)] = imgs.var(axis=0, ddof=1)
np.savez_compressed("gpurun_out/r03_sampler_dump.npz", **out)
print("saved", list(out))
