import os, sys
sys.path.insert(0, '/root/repo')
import torch
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip
fs = load_scene('tests/golden/c3_random4d.ndtscene.gz')
g = NdtHip(0); g.upload_scene(fs)
for i in range(3): g.render(1920, 1080, 4)
os.environ['NDT_HIP_DEBUG_LEVELS'] = '1'
out, st = g.render(1920, 1080, 4, profile=1)
print(st.as_dict())
