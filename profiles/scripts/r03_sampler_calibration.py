"""The statistic tests/test_gpu_parity.py::test_jittered_samples_statistically_match_the_oracle applies to the device's image,
applied to the ORACLE's own images: every one of 25 oracle renders (different drand48 states) against the ensemble of the
other 24.  If the per-value z of an independent draw from the SAME distribution has a mean that is not zero -- skewed pixel
distributions, |z| clipped at 15 -- the device's all-negative `mean z` of round 2 is a property of the statistic, not of
the sampler.  CPU only:  python profiles/scripts/r03_sampler_calibration.py [case ...]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from conftest import golden, SAMPLED_CASES, Oracle

oracle = Oracle()
names = sys.argv[1:] or SAMPLED_CASES
for name in names:
    g = golden(name)
    s0 = g.meta["seed48"]
    for S in (8, 128):
        imgs = []
        for k in range(25):
            img, _ = oracle.render(g.scene, g.width, g.height, g.depth, samples=S, seed48=[(s0[0] + 7919 * k) & 0xffff, s0[1], s0[2]],
                                   stereo=g.meta.get("stereo", 0))
            imgs.append(img)
        imgs = np.array(imgs)
        means, fracs, beyond = [], [], []
        for k in range(25):
            ens = np.delete(imgs, k, axis=0)
            mu, sd = ens.mean(axis=0), ens.std(axis=0, ddof=1)
            noisy = sd > 1e-12
            z = (imgs[k] - mu)[noisy] / (sd[noisy] * np.sqrt(1.0 + 1.0 / 24))
            means.append(np.clip(z, -15, 15).mean())
            fracs.append((np.abs(z) < 5).mean())
            beyond.append(int((np.abs(z) >= 15).sum()))
        means = np.array(means)
        print("%s S=%d: held-out oracle draws: mean z %+.3f +- %.3f (min %+.3f, max %+.3f; %d of 25 negative), |z|<5 %.2f %% (worst %.2f %%), beyond 15: %s" % (
            name, S, means.mean(), means.std(ddof=1), means.min(), means.max(), int((means < 0).sum()), 100 * np.mean(fracs), 100 * min(fracs),
            sorted(beyond)[-3:]), flush=True)
