"""Per-launch timing probe for the trace kernel (development aid, not a test).
usage: python profiles/levels_probe.py [--drop-type hcube] [--scene c3_random4d] [--res 1920x1080] [--depth 4]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from ndt_amd import load_scene, OBJ_TYPES
from ndt_amd.hip import NdtHip

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3_random4d")
ap.add_argument("--res", default="1920x1080")
ap.add_argument("--depth", type=int, default=4)
ap.add_argument("--drop-type", default="")
ap.add_argument("--aa", default="", help="diff,depth: recursive anti-aliasing (every pass prints)")
ap.add_argument("--opt", action="append", default=[], help="name=value for ndt_hip_set_option, before the scene is uploaded")
a = ap.parse_args()
fs = load_scene("tests/golden/%s.ndtscene.gz" % a.scene)
if a.drop_type:
    t = OBJ_TYPES.index(a.drop_type)
    drop = {i for i, o in enumerate(fs.objects) if o["type"] == t and o["parent"] < 0}
    refs = []
    for k in fs.kd_nodes:
        if k["dim"] < 0:
            ids = [i for i in fs.leaf_refs[k["first"]:k["first"] + k["num"]] if i not in drop]
            k["first"], k["num"] = len(refs), len(ids)
            refs.extend(ids)
    fs.leaf_refs = refs
    fs._struct = None
    fs.finalize()
    print("dropped %d objects of type %s from the leaves" % (len(drop), a.drop_type))
w, h = (int(x) for x in a.res.split("x"))
g = NdtHip(0)
g.set_option("pipeline", 1)             # the per-bounce kernels: what the phase stamps instrument
for kv in a.opt:
    g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
g.upload_scene(fs)
for i in range(3 if not a.aa else 1):
    g.render(w, h, a.depth)
g.set_option("debug_levels", 1)
aa = tuple(int(x) for x in a.aa.split(",")) if a.aa else None
out, st = g.render(w, h, a.depth, profile=1, aa=aa)
print(st.as_dict())
