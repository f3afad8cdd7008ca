#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the COMPILED REFERENCE.

Run in the build container only (needs /root/reference):

    make -C oracle ref          # compiles the reference where it lies -> oracle/_ref/
    python tests/golden/make_golden.py [case ...]

For every case this drives oracle/_ref/ndt_ref_shim (oracle/ref_shim.c) to
  * flatten the prepared scene the reference built          -> <case>.ndtscene.gz
  * render it with the reference's own render_image          -> <case>.npz : fb (H,W,4) float64,
                                                                or rgba8 (H,W,4) uint8 for the
                                                                full-resolution cases (pixel_d2c)
  * answer seeded trace_kd queries                           -> <case>.npz : kat_in, kat_out
  * count trace_kd calls during the render                   -> <case>.json
The fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import gzip
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ndt_amd import load_scene  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(REF, "ndt_ref_shim")

# name -> scene .so, dims, WxH, -l depth, what to keep
CASES = {
    # BASELINE.json configs[0] at reduced size: double framebuffer
    "c1_hypercube3d": dict(scene="hypercube", dims=3, res=(96, 96), depth=128, fb=True, kat=2048),
    # configs[1], configs[2] at reduced size: double framebuffers + known-answer rays
    "c2_balls4d": dict(scene="balls", dims=4, res=(128, 72), depth=128, fb=True, kat=4096),
    "c3_random4d": dict(scene="random", dims=4, res=(128, 72), depth=4, fb=True, kat=4096),
    # configs[4] family at reduced size; 5-D exercises the odd-N lane-pair dot order
    "c5_hypercube4d": dict(scene="hypercube", dims=4, res=(64, 36), depth=128, fb=True, kat=1024),
    "c5_hypercube5d": dict(scene="hypercube", dims=5, res=(48, 27), depth=128, fb=True, kat=1024),
    # configs[4]: 6-D .. 8-D hypercubes (728 / 2186 / 6560 objects): scenes too big for LDS and for
    # a register visit mask, so these pin the global-memory tier of the trace kernel
    "c5_hypercube6d": dict(scene="hypercube", dims=6, res=(48, 27), depth=128, fb=True, kat=512),
    "c5_hypercube7d": dict(scene="hypercube", dims=7, res=(32, 18), depth=128, fb=True, kat=256),
    "c5_hypercube8d": dict(scene="hypercube", dims=8, res=(24, 14), depth=128, fb=True, kat=128),
    # beyond BASELINE's sweep (the reference takes any -d >= 3, ndt.c:1499-1503): 9-D and 10-D.  The zoo (every type; its
    # hcube has 16 866 / 52 904 nested faces there) and the 9-D hypercube (19 682 objects, 1025 kd nodes).  (random.c's four
    # hcubes make a 12 MB / 42 MB scene file in 9-D / 10-D and its camera sees none of it: not kept.)
    "zoo9d": dict(scene="parity_zoo", dims=9, res=(40, 24), depth=5, fb=True, kat=512),
    "zoo10d": dict(scene="parity_zoo", dims=10, res=(40, 24), depth=5, fb=True, kat=512),
    # 11-D and 12-D: the zoo without its hcube (164 k / 500 k nested faces there)
    "zoo11d": dict(scene="parity_zoo", dims=11, res=(40, 24), depth=5, fb=True, kat=512, config="nohcube"),
    "zoo12d": dict(scene="parity_zoo", dims=12, res=(40, 24), depth=5, fb=True, kat=512, config="nohcube"),
    "c5_hypercube9d": dict(scene="hypercube", dims=9, res=(24, 14), depth=128, fb=True, kat=1024, kat_aimed=True),
    # tests/scenes/parity_zoo.c (this repo's own scene program, compiled against the reference):
    # spot light, LIGHT_AMBIENT entry + scn->ambient, glass with total internal reflection, finite
    # hcylinder, infinite cylinder, computed-normal hfacet, skewed hcube, rotated cluster
    "zoo4d": dict(scene="parity_zoo", dims=4, res=(96, 54), depth=6, fb=True, kat=2048),
    "zoo3d_mirror": dict(scene="parity_zoo", dims=3, res=(64, 48), depth=128, fb=True, kat=512, config="mirror"),
    "zoo5d_f2": dict(scene="parity_zoo", dims=5, res=(48, 27), depth=8, fb=True, kat=512, frame=2),
    "zoo6d": dict(scene="parity_zoo", dims=6, res=(40, 24), depth=5, fb=True, kat=256),
    # an animated frame: rotated hypercube, different tree
    "c1_hypercube3d_f37": dict(scene="hypercube", dims=3, res=(64, 64), depth=128, fb=True, kat=0, frame=37),
    # full BASELINE resolution, 8-bit (what the reference writes to PNG)
    "c2_balls4d_1080p": dict(scene="balls", dims=4, res=(1920, 1080), depth=128, fb=False, rgba8=True, kat=0,
                             share_scene="c2_balls4d"),
    "c3_random4d_1080p": dict(scene="random", dims=4, res=(1920, 1080), depth=4, fb=False, rgba8=True, kat=0,
                              share_scene="c3_random4d"),
    # every BASELINE config at its stated size (8-bit, like the two above): configs[0] 256x256, configs[3]'s 3840x2160
    # frame, configs[4]'s 6-D .. 8-D sweep at 1920x1080
    "c1_hypercube3d_256": dict(scene="hypercube", dims=3, res=(256, 256), depth=128, fb=False, rgba8=True, kat=0,
                               share_scene="c1_hypercube3d"),
    "c4_random4d_4k": dict(scene="random", dims=4, res=(3840, 2160), depth=4, fb=False, rgba8=True, kat=0,
                           share_scene="c3_random4d"),
    "c5_hypercube6d_1080p": dict(scene="hypercube", dims=6, res=(1920, 1080), depth=128, fb=False, rgba8=True, kat=0,
                                 share_scene="c5_hypercube6d"),
    "c5_hypercube7d_1080p": dict(scene="hypercube", dims=7, res=(1920, 1080), depth=128, fb=False, rgba8=True, kat=0,
                                 share_scene="c5_hypercube7d"),
    "c5_hypercube8d_1080p": dict(scene="hypercube", dims=8, res=(1920, 1080), depth=128, fb=False, rgba8=True, kat=0,
                                 share_scene="c5_hypercube8d"),
    # known answers only (an 8x8 render that is not kept): 8192 trace_kd queries each for the scenes of the
    # global-memory tier, half of them aimed at the items of every kd leaf (make_kat_rays, aimed=True)
    "kat_hypercube6d": dict(scene="hypercube", dims=6, res=(8, 8), depth=2, fb=False, kat=8192, kat_aimed=True,
                            share_scene="c5_hypercube6d"),
    "kat_hypercube7d": dict(scene="hypercube", dims=7, res=(8, 8), depth=2, fb=False, kat=8192, kat_aimed=True,
                            share_scene="c5_hypercube7d"),
    "kat_hypercube8d": dict(scene="hypercube", dims=8, res=(8, 8), depth=2, fb=False, kat=8192, kat_aimed=True,
                            share_scene="c5_hypercube8d"),
    # Whitted's recursive anti-aliasing (-a diff,depth): fb = the resampled image in doubles, produced by the
    # reference's own render_line + resample_pixel (the shim's --aa mode); reference defaults are 20,4
    "aa_c3_random4d": dict(scene="random", dims=4, res=(64, 36), depth=4, fb=True, kat=0, aa=(20, 4),
                           share_scene="c3_random4d"),
    "aa_c1_hypercube3d": dict(scene="hypercube", dims=3, res=(48, 48), depth=128, fb=True, kat=0, aa=(20, 2),
                              share_scene="c1_hypercube3d"),
    "aa_zoo4d": dict(scene="parity_zoo", dims=4, res=(48, 27), depth=6, fb=True, kat=0, aa=(8, 3),
                     share_scene="zoo4d"),
    # ... with the stereo modes that split the image, and through the VR camera (every sample goes through render_pixel)
    "aa_zoo4d_sbs": dict(scene="parity_zoo", dims=4, res=(64, 36), depth=6, fb=True, kat=0, aa=(8, 3), stereo=1, v2=True,
                         share_scene="st_zoo4d_sbs"),
    "aa_zoo4d_ou": dict(scene="parity_zoo", dims=4, res=(48, 54), depth=6, fb=True, kat=0, aa=(8, 2), stereo=2, v2=True,
                        share_scene="st_zoo4d_sbs"),
    "aa_vr_zoo4d": dict(scene="parity_zoo", dims=4, res=(64, 36), depth=6, fb=True, kat=0, aa=(8, 3), v2=True, config="vr",
                        share_scene="vr_zoo4d"),
    # stereo modes (-m, ndt.c:46-48, 590-650), depth maps (-z, ndt.c:362-373, 753-756), VR / panorama cameras
    # (camera.c:506-555): scenes in `ndtscene 2` (eyes, local axes, fields of view)
    "st_zoo4d_sbs": dict(scene="parity_zoo", dims=4, res=(64, 36), depth=6, fb=True, kat=0, stereo=1, v2=True),
    "st_zoo4d_ou": dict(scene="parity_zoo", dims=4, res=(48, 54), depth=6, fb=True, kat=0, stereo=2, v2=True,
                        share_scene="st_zoo4d_sbs"),
    "st_zoo3d_anaglyph": dict(scene="parity_zoo", dims=3, res=(48, 36), depth=6, fb=True, kat=0, stereo=3, v2=True,
                              depth_map=True),
    # HIDEF_3D frame packing: 1080 lines left eye, 45 blank, 1080 right eye (ndt.c:614-631); narrow to stay small
    "st_zoo3d_hidef": dict(scene="parity_zoo", dims=3, res=(24, 2205), depth=4, fb=True, kat=0, stereo=4, v2=True,
                           share_scene="st_zoo3d_anaglyph"),
    "vr_zoo4d": dict(scene="parity_zoo", dims=4, res=(64, 36), depth=6, fb=True, kat=0, v2=True, config="vr",
                     depth_map=True),
    "pano_zoo5d_sbs": dict(scene="parity_zoo", dims=5, res=(64, 32), depth=5, fb=True, kat=0, v2=True, config="pano",
                           stereo=1),
    "depth_c3_random4d": dict(scene="random", dims=4, res=(64, 36), depth=4, fb=True, kat=0, v2=True, depth_map=True),
    # -n samples > 1: jittered samples + lens sampling from the global drand48 stream (ndt.c:505-542), one thread,
    # fresh process (default seed): reproducible, and the oracle walks the same stream
    "ns_c3_random4d": dict(scene="random", dims=4, res=(32, 18), depth=4, fb=True, kat=0, samples=4, share_scene="c3_random4d"),
    "ns_zoo4d_dof": dict(scene="parity_zoo", dims=4, res=(32, 18), depth=6, fb=True, kat=0, samples=6, v2=True, config="dof"),
    # ... in a side-by-side image (the jitter is added after render_pixel's split) and through the VR camera
    "ns_zoo4d_sbs": dict(scene="parity_zoo", dims=4, res=(32, 18), depth=6, fb=True, kat=0, samples=4, stereo=1, v2=True,
                         share_scene="st_zoo4d_sbs"),
    "ns_vr_zoo4d": dict(scene="parity_zoo", dims=4, res=(32, 18), depth=6, fb=True, kat=0, samples=4, v2=True, config="vr",
                        share_scene="vr_zoo4d"),
    # ... as an anaglyph (two adaptive loops per pixel, left eye then right, ndt.c:636-647), frame-packed (the jitter spans
    # 1/height of the 1080-line eye image, ndt.c:482-483 against :629), and with a depth map (the LAST sample's hit, ndt.c:362-373)
    "ns_zoo3d_anaglyph": dict(scene="parity_zoo", dims=3, res=(32, 18), depth=6, fb=True, kat=0, samples=4, stereo=3, v2=True,
                              depth_map=True, share_scene="st_zoo3d_anaglyph"),
    "ns_zoo3d_hidef": dict(scene="parity_zoo", dims=3, res=(12, 2205), depth=4, fb=True, kat=0, samples=3, stereo=4, v2=True,
                           share_scene="st_zoo3d_anaglyph"),
    "ns_c3_random4d_depth": dict(scene="random", dims=4, res=(32, 18), depth=4, fb=True, kat=0, samples=4, v2=True, depth_map=True,
                                 share_scene="depth_c3_random4d"),
    # recursive anti-aliasing as an anaglyph (every sample is the mix of two eyes; the subdivision tests see the mix), with a
    # depth map (the first pass's), and with a lens (every sample draws its lens point: a stochastic render, ndt.c:528)
    "aa_zoo3d_anaglyph": dict(scene="parity_zoo", dims=3, res=(48, 36), depth=6, fb=True, kat=0, aa=(8, 2), stereo=3, v2=True,
                              share_scene="st_zoo3d_anaglyph"),
    "aa_c3_random4d_depth": dict(scene="random", dims=4, res=(64, 36), depth=4, fb=True, kat=0, aa=(20, 3), v2=True, depth_map=True,
                                 share_scene="depth_c3_random4d"),
    "aa_zoo4d_dof": dict(scene="parity_zoo", dims=4, res=(32, 18), depth=6, fb=True, kat=0, aa=(8, 2), v2=True, config="dof",
                         share_scene="ns_zoo4d_dof"),
    # area lights (LIGHT_DISK / LIGHT_RECT, ndt.c:116-147): a random point of the light per shading evaluation, so even
    # -n 1 is stochastic (the adaptive loop's repeats differ); the second case adds jitter and a lens
    "al_zoo4d": dict(scene="parity_zoo", dims=4, res=(32, 18), depth=5, fb=True, kat=0, samples=1, v2=True, config="area"),
    "al_zoo3d_dof_n3": dict(scene="parity_zoo", dims=3, res=(32, 18), depth=5, fb=True, kat=0, samples=3, v2=True,
                            config="area,dof"),
}


def run_shim(args):
    cmd = [SHIM, "--objects", os.path.join(REF, "objects")] + args
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    info = {}
    m = re.search(r"ref_shim: render_s ([0-9.]+) threads (\d+)", out)
    if m:
        info["ref_render_s"] = float(m.group(1))
        info["ref_threads"] = int(m.group(2))
    m = re.search(r"ref_shim: aa_diff (\d+) aa_depth (\d+) pixels_resampled (\d+) rays_pass1 (\d+)", out)
    if m:
        info["aa_diff"], info["aa_depth"], info["pixels_resampled"], info["rays_pass1"] = (int(m.group(i)) for i in (1, 2, 3, 4))
    m = re.search(r"ref_shim: seed48 (\d+) (\d+) (\d+)", out)
    if m:
        info["seed48"] = [int(m.group(i)) for i in (1, 2, 3)]
    m = re.search(r"ref_shim: rays_closest (\d+) rays_shadow (\d+) rays_total (\d+)", out)
    if m:
        info["rays_closest"], info["rays_shadow"], info["rays_total"] = (int(m.group(i)) for i in (1, 2, 3))
    return info


def leaf_targets(fs, rng, n):
    """n points inside the bounding spheres of kd-leaf items, walking the leaves round-robin so that every leaf
    (and, as n allows, every item of it) is aimed at."""
    leaves = [k for k in fs.kd_nodes if k["dim"] < 0 and k["num"] > 0]
    out = []
    turn = 0
    while len(out) < n and leaves:
        for k in leaves:
            if len(out) >= n:
                break
            item = fs.leaf_refs[k["first"] + turn % k["num"]]
            o = fs.objects[item]
            c = fs.vec(o["bounds_center_off"])
            r = o["bounds_radius"] if o["bounds_radius"] > 0 else 0.5
            u = rng.standard_normal(fs.dims)
            u /= np.linalg.norm(u)
            out.append(c + u * r * rng.random() ** (1.0 / fs.dims))
        turn += 1
    return out


def make_kat_rays(fs, n, seed, aimed=False):
    """Seeded query rays: camera-like, interior random, and shadow-like with limits.  aimed=True: the second half of
    the rays go for points inside the bounding spheres of the items of every kd leaf (same four kinds of ray)."""
    if aimed:
        head = make_kat_rays(fs, n // 2, seed)
        rng = np.random.default_rng(seed + 77)
        d = fs.dims
        cam = fs.vec(fs.cam["pos"])
        lo, hi = fs.vec(fs.bb["lower"]), fs.vec(fs.bb["upper"])
        if not np.all(np.isfinite(lo)) or np.any(lo > hi):
            lo, hi = -10 * np.ones(d), 10 * np.ones(d)
        tail = np.zeros((n - n // 2, 2 * d + 1))
        for i, tgt in enumerate(leaf_targets(fs, rng, n - n // 2)):
            kind = i % 4
            o = cam if kind == 0 else lo - 1 + rng.random(d) * (hi - lo + 2)
            v = tgt - o
            nv = np.linalg.norm(v)
            if nv < 1e-9:
                v, nv = np.ones(d), np.sqrt(d)
            lim = -1.0 if kind < 2 else (0.0 if kind == 2 else float(nv) * rng.uniform(0.3, 1.2))
            tail[i, :d] = o
            tail[i, d:2 * d] = v / nv
            tail[i, 2 * d] = lim
        return np.concatenate([head, tail])
    rng = np.random.default_rng(seed)
    d = fs.dims
    cam = fs.vec(fs.cam["pos"])
    lo, hi = fs.vec(fs.bb["lower"]), fs.vec(fs.bb["upper"])
    if not np.all(np.isfinite(lo)) or np.any(lo > hi):
        lo, hi = -10 * np.ones(d), 10 * np.ones(d)
    rays = np.zeros((n, 2 * d + 1))
    for i in range(n):
        kind = i % 4
        if kind == 0:      # from the camera into the scene box
            o = cam
            tgt = lo + rng.random(d) * (hi - lo)
            v = tgt - o
            lim = -1.0
        elif kind == 1:    # between two random points of the (slightly grown) box
            o = lo - 1 + rng.random(d) * (hi - lo + 2)
            tgt = lo + rng.random(d) * (hi - lo)
            v = tgt - o
            lim = -1.0
        elif kind == 2:    # any-hit (directional-light shadow rays use limit 0, ndt.c:184)
            o = lo + rng.random(d) * (hi - lo)
            v = rng.standard_normal(d)
            lim = 0.0
        else:              # point-light style: positive limit (ndt.c:187-188)
            o = lo - 2 + rng.random(d) * (hi - lo + 4)
            tgt = lo + rng.random(d) * (hi - lo)
            v = tgt - o
            lim = float(np.linalg.norm(v)) * rng.uniform(0.3, 1.2)
        if i % 16 == 5:    # axis-parallel components exercise the v_inv clamp (kd-tree.c:583-588)
            v = v.copy()
            v[rng.integers(0, d)] = 0.0
        nv = np.linalg.norm(v)
        if nv < 1e-9:
            v = np.ones(d)
            nv = np.linalg.norm(v)
        rays[i, :d] = o
        rays[i, d:2 * d] = v / nv
        rays[i, 2 * d] = lim
    return rays


def generate(name, case):
    print("==", name, flush=True)
    w, h = case["res"]
    frame = case.get("frame", 0)
    threads = str(os.cpu_count() or 1)
    base = ["--scene", os.path.join(REF, "scenes", case["scene"] + ".so"), "--dims", str(case["dims"]),
            "--frame", str(frame), "--res", "%dx%d" % (w, h), "--threads", threads, "--depth", str(case["depth"])]
    if case.get("config"):
        base += ["--config", case["config"]]
    meta = dict(name=name, scene=case["scene"], dims=case["dims"], width=w, height=h, depth=case["depth"],
                frame=frame, config=case.get("config"), generator="tests/golden/make_golden.py via oracle/ref_shim.c")
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        scene_txt = os.path.join(tmp, "scene.txt")
        # pass 1: scene only
        run_shim(base + ["--tmp", tmp, "--scene-out", scene_txt, "--no-render"] + (["--scene-v2"] if case.get("v2") else []))
        fs = load_scene(scene_txt)
        share = case.get("share_scene")
        if share:
            # same scene as another case: verify and reference it instead of storing a copy
            with gzip.open(os.path.join(HERE, share + ".ndtscene.gz"), "rt") as g:
                if g.read() != open(scene_txt).read():
                    raise SystemExit("scene of %s differs from %s" % (name, share))
            meta["scene_file"] = share + ".ndtscene.gz"
        else:
            with open(scene_txt, "rb") as src, open(os.path.join(HERE, name + ".ndtscene.gz"), "wb") as raw:
                with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as dst:
                    dst.write(src.read())
            meta["scene_file"] = name + ".ndtscene.gz"
        meta["objects"] = fs.type_histogram()
        meta["kd_nodes"] = len(fs.kd_nodes)
        # pass 2: render (+ known answers)
        args = base + ["--tmp", tmp, "--fb-out", os.path.join(tmp, "fb.bin")]
        if case.get("aa"):
            args += ["--aa", "%d,%d" % case["aa"]]
        if case.get("stereo"):
            args += ["--stereo", str(case["stereo"])]
            meta["stereo"] = case["stereo"]
        if case.get("depth_map"):
            args += ["--depth-out", os.path.join(tmp, "depth.bin")]
        if case.get("samples"):
            args += ["--samples", str(case["samples"])]
            args[args.index("--threads") + 1] = "1"
            meta["samples"] = case["samples"]
        if case["kat"]:
            rays = make_kat_rays(fs, case["kat"], seed=1234 + case["dims"], aimed=bool(case.get("kat_aimed")))
            rays.tofile(os.path.join(tmp, "rays.bin"))
            args += ["--rays-in", os.path.join(tmp, "rays.bin"), "--rays-out", os.path.join(tmp, "kat.bin")]
        meta.update(run_shim(args))
        fb = np.fromfile(os.path.join(tmp, "fb.bin")).reshape(h, w, 4)
        if case.get("fb"):
            arrays["fb"] = fb
        if case.get("depth_map"):
            arrays["depth"] = np.fromfile(os.path.join(tmp, "depth.bin")).reshape(h, w)
        if case.get("rgba8"):
            # pixel_d2c (image.h:36-39): the byte the reference's PNG/JPEG writer stores
            arrays["rgba8"] = (np.sqrt(np.maximum(0.0, np.minimum(1.0, fb))) * 255).astype(np.uint8)
            meta["fb_max"] = float(fb[..., :3].max())
        if case["kat"]:
            d = case["dims"]
            arrays["kat_in"] = rays
            arrays["kat_out"] = np.fromfile(os.path.join(tmp, "kat.bin")).reshape(-1, 2 + 2 * d)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
        f.write("\n")
    print("   ", {k: v for k, v in meta.items() if k.startswith("rays") or k.startswith("ref_")}, flush=True)


# YAML scene files (`ndt -y` writes them, `-s scenes/yaml.so -u file.yaml` reads them; scene.c:573-2177).
# name -> the frames of which scene program the reference writes into one file (one document per
# frame), and which document it then loads back through its own scenes/yaml.so.  Stored under
# tests/golden/yaml/: <name>.yaml.gz (what the reference wrote) and <name>.ndtscene.gz (the scene
# the reference built from it, flattened) (+ <name>.npz with the framebuffer it rendered, if asked).
YAML_CASES = {
    "y_random4d": dict(scene="random", dims=4, frames=[0], load=0, render=(64, 36), depth=4),
    "y_hypercube3d_2frames": dict(scene="hypercube", dims=3, frames=[0, 37], load=1),
    "y_balls4d": dict(scene="balls", dims=4, frames=[0], load=0),
    "y_zoo5d": dict(scene="parity_zoo", dims=5, frames=[2], load=0),
    "y_zoo3d_mirror": dict(scene="parity_zoo", dims=3, frames=[0], load=0, config="mirror"),
    "y_hypercube6d": dict(scene="hypercube", dims=6, frames=[0], load=0),
}


def generate_yaml(name, case):
    print("==", name, flush=True)
    out_dir = os.path.join(HERE, "yaml")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        text = ""
        for fr in case["frames"]:
            part = os.path.join(tmp, "f%d.yaml" % fr)
            args = ["--scene", os.path.join(REF, "scenes", case["scene"] + ".so"), "--dims", str(case["dims"]),
                    "--frame", str(fr), "--res", "8x8", "--no-render", "--yaml-out", part, "--tmp", tmp]
            if case.get("config"):
                args += ["--config", case["config"]]
            run_shim(args)
            text += open(part).read()
        yaml_path = os.path.join(tmp, name + ".yaml")
        with open(yaml_path, "w") as f:
            f.write(text)
        scene_txt = os.path.join(tmp, "scene.txt")
        base = ["--scene", os.path.join(REF, "scenes", "yaml.so"), "--config", yaml_path, "--dims", str(case["dims"]),
                "--frame", str(case["load"]), "--tmp", tmp]
        run_shim(base + ["--res", "8x8", "--no-render", "--scene-out", scene_txt])
        for src, dst in ((yaml_path, name + ".yaml.gz"), (scene_txt, name + ".ndtscene.gz")):
            with open(src, "rb") as fi, open(os.path.join(out_dir, dst), "wb") as raw:
                with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as fo:
                    fo.write(fi.read())
        meta = dict(name=name, scene=case["scene"], dims=case["dims"], frames_written=case["frames"], frame_loaded=case["load"],
                    config=case.get("config"), generator="tests/golden/make_golden.py via oracle/ref_shim.c")
        if case.get("render"):
            w, h = case["render"]
            info = run_shim(base + ["--res", "%dx%d" % (w, h), "--depth", str(case["depth"]), "--threads", str(os.cpu_count() or 1),
                                    "--fb-out", os.path.join(tmp, "fb.bin")])
            meta.update(info)
            meta.update(width=w, height=h, depth=case["depth"])
            np.savez_compressed(os.path.join(out_dir, name + ".npz"), fb=np.fromfile(os.path.join(tmp, "fb.bin")).reshape(h, w, 4))
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
            f.write("\n")


def main():
    if not os.path.exists(SHIM):
        raise SystemExit("build the reference first: make -C oracle ref")
    names = sys.argv[1:] or (list(CASES) + list(YAML_CASES))
    for name in names:
        if name in YAML_CASES:
            generate_yaml(name, YAML_CASES[name])
        else:
            generate(name, CASES[name])


if __name__ == "__main__":
    main()
