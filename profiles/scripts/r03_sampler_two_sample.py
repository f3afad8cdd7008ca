"""Device ensemble against oracle ensemble (stochastic paths): K device renders with different "sample_seed" values and K oracle
renders from different drand48 states are two samples of -- if the device's sampler is right -- one distribution.  Per value:
difference of the ensemble means over its standard error; overall: mean of that t over the noisy values (0 +- 1/sqrt(n) if
unbiased), and the image means.  GPU:  python profiles/scripts/r03_sampler_two_sample.py [case ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from conftest import golden, SAMPLED_CASES, Oracle
from ndt_amd.hip import NdtHip

oracle = Oracle()
gpu = NdtHip(0)
K = 48
names = sys.argv[1:] or SAMPLED_CASES
for name in names:
    g = golden(name)
    s0 = g.meta["seed48"]
    gpu.upload_scene(g.scene)
    stereo = g.meta.get("stereo", 0)
    for S in (8, 64):
        ora, dev, taken_o, taken_d = [], [], [], []
        for k in range(K):
            img, so = oracle.render(g.scene, g.width, g.height, g.depth, samples=S, seed48=[(s0[0] + 7919 * k) & 0xffff, s0[1], s0[2]], stereo=stereo)
            ora.append(img)
            taken_o.append(so.rays_primary)
            gpu.set_option("sample_seed", k)
            img, st = gpu.render(g.width, g.height, g.depth, samples=S, stereo=stereo)
            dev.append(img)
            taken_d.append(st.aa_samples)
        gpu.set_option("sample_seed", 0)
        ora, dev = np.array(ora), np.array(dev)
        mo, md = ora.mean(axis=0), dev.mean(axis=0)
        se = np.sqrt(ora.var(axis=0, ddof=1) / K + dev.var(axis=0, ddof=1) / K)
        noisy = se > 1e-12
        t = (md - mo)[noisy] / se[noisy]
        n = int(noisy.sum())
        io, idv = ora[..., :3].mean(axis=(1, 2, 3)), dev[..., :3].mean(axis=(1, 2, 3))
        t_img = (idv.mean() - io.mean()) / np.sqrt(io.var(ddof=1) / K + idv.var(ddof=1) / K)
        print("%s S=%d: %d noisy values; per-value t (device - oracle): mean %+.3f (unbiased: 0 +- %.3f), rms %.3f (unbiased: ~1), |t|>4: %d; "
              "image mean device %.6f oracle %.6f (t %+.2f); samples per pixel device %.2f oracle %.2f" % (
                  name, S, n, t.mean(), 1 / np.sqrt(n), np.sqrt((t ** 2).mean()), int((np.abs(t) > 4).sum()), idv.mean(), io.mean(), t_img,
                  np.mean(taken_d) / (g.width * g.height), np.mean(taken_o) / (g.width * g.height)), flush=True)
        worst = np.argsort(-np.abs((md - mo) / np.where(se > 1e-12, se, 1e30)).reshape(-1))[:5]
        for w in worst:
            idx = np.unravel_index(w, md.shape)
            print("     value %s: device %.6f oracle %.6f se %.2g" % (idx, md[idx], mo[idx], se[idx]))
gpu.close()
