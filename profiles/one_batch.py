"""Latency of a single 64-ray batch of k_trace (development aid): run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
fs = load_scene("tests/golden/c3_random4d.ndtscene.gz")
z = np.load("tests/golden/c3_random4d.npz")
rays = z["kat_in"]
g = NdtHip(0); g.upload_scene(fs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
closest = rays[rays[:, 8] < 0]
for i in range(12):
    g.trace_rays(closest[:n])
