import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from ndt_amd import load_scene, RenderParams, RenderStats  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Oracle:
    """ctypes face of oracle/libndt_oracle.so -- the CPU checker (test infrastructure only)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "libndt_oracle.so")
        src = os.path.join(ROOT, "oracle", "ndt_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, capture_output=True)
        self.lib = C.CDLL(path)
        self.lib.ndt_oracle_render.restype = C.c_int
        self.lib.ndt_oracle_render_depth.restype = C.c_int
        self.lib.ndt_oracle_trace_rays.restype = C.c_int

    def render(self, fs, width, height, depth, row_begin=0, row_step=1, threads=None, literal=False, specular=1, aa=None,
               stereo=0, depth_map=False, samples=1, seed48=None):
        """aa = (aa_diff, aa_depth) switches Whitted's recursive anti-aliasing on (-a diff,depth);
        stereo = ndt_stereo_mode; depth_map=True also returns the 1/distance map: (rgba, depth, stats)."""
        from ndt_amd import shard_rows
        rows = shard_rows(height, row_begin, row_step)
        p = RenderParams(width, height, depth, 1, row_begin, row_step, specular, 0)
        if aa is not None:
            p.recursive_aa, p.aa_diff, p.aa_depth = 1, aa[0], aa[1]
        p.stereo = stereo
        p.samples = samples
        if seed48 is not None:
            self.lib.ndt_oracle_set_seed48(C.c_ushort(seed48[0]), C.c_ushort(seed48[1]), C.c_ushort(seed48[2]))
        st = RenderStats()
        out = np.zeros((rows, width, 4), dtype=np.float64)
        dm = np.zeros((rows, width), dtype=np.float64) if depth_map else None
        threads = threads or min(8, os.cpu_count() or 1)
        rc = self.lib.ndt_oracle_render_depth(fs.byref(), C.byref(p), out.ctypes.data_as(C.c_void_p),
                                              dm.ctypes.data_as(C.c_void_p) if depth_map else None, C.byref(st),
                                              C.c_int(threads), C.c_int(1 if literal else 0))
        assert rc == 0, "oracle render failed rc=%d" % rc
        return (out, dm, st) if depth_map else (out, st)

    def trace(self, fs, rays):
        d = fs.dims
        n = rays.shape[0]
        o = np.ascontiguousarray(rays[:, :d])
        v = np.ascontiguousarray(rays[:, d:2 * d])
        lim = np.ascontiguousarray(rays[:, 2 * d])
        obj = np.zeros(n, dtype=np.int32)
        hit = np.zeros((n, d))
        nrm = np.zeros((n, d))
        rc = self.lib.ndt_oracle_trace_rays(fs.byref(), C.c_int64(n), o.ctypes.data_as(C.c_void_p),
                                            v.ctypes.data_as(C.c_void_p), lim.ctypes.data_as(C.c_void_p),
                                            obj.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p),
                                            nrm.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return obj, hit, nrm

    def quantize(self, rgba):
        flat = np.ascontiguousarray(rgba, dtype=np.float64)
        out = np.zeros(flat.shape, dtype=np.uint8)
        self.lib.ndt_oracle_quantize(flat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                     C.c_int64(flat.size))
        return out


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


class Golden:
    def __init__(self, name):
        self.name = name
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.meta = json.load(f)
        self.scene = load_scene(os.path.join(GOLDEN, self.meta["scene_file"]))
        self.data = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.width, self.height, self.depth = self.meta["width"], self.meta["height"], self.meta["depth"]


def comparable(fb, stereo):
    """What of a framebuffer the reference defines: with HIDEF_3D frame packing the 45 blank lines between the
    eyes get r = g = b = 0 and an alpha that is never set (ndt.c:624-627); those alphas are masked out."""
    if stereo != 4:
        return fb
    out = np.array(fb, copy=True)
    out[1080:1080 + 46, :, 3] = 0.0
    return out


_golden_cache = {}


def golden(name):
    if name not in _golden_cache:
        _golden_cache[name] = Golden(name)
    return _golden_cache[name]


SMALL_CASES = ["c1_hypercube3d", "c1_hypercube3d_f37", "c2_balls4d", "c3_random4d", "c5_hypercube4d",
               "c5_hypercube5d", "c5_hypercube6d", "c5_hypercube7d", "c5_hypercube8d", "zoo4d", "zoo3d_mirror",
               "zoo5d_f2", "zoo6d", "zoo9d", "zoo10d", "zoo11d", "zoo12d", "c5_hypercube9d"]
KAT_CASES = ["c1_hypercube3d", "c2_balls4d", "c3_random4d", "c5_hypercube4d", "c5_hypercube5d", "c5_hypercube6d",
             "c5_hypercube7d", "c5_hypercube8d", "zoo4d", "zoo3d_mirror", "zoo5d_f2", "zoo6d", "zoo9d", "zoo10d", "zoo11d", "zoo12d", "c5_hypercube9d",
             # 8192 queries each, half of them aimed at the items of every kd leaf: the global-memory tier
             "kat_hypercube6d", "kat_hypercube7d", "kat_hypercube8d"]
# every BASELINE config at its stated size, 8-bit like the reference's PNG: configs[0] 256x256, [1] and [2] 1920x1080,
# [3]'s 3840x2160 frame, [4]'s 6-D .. 8-D sweep at 1920x1080
FULL_CASES = ["c1_hypercube3d_256", "c2_balls4d_1080p", "c3_random4d_1080p", "c4_random4d_4k", "c5_hypercube6d_1080p",
              "c5_hypercube7d_1080p", "c5_hypercube8d_1080p"]
AA_CASES = ["aa_c3_random4d", "aa_c1_hypercube3d", "aa_zoo4d", "aa_zoo4d_sbs", "aa_zoo4d_ou", "aa_vr_zoo4d",
            # as an anaglyph (every sample is the mix of two eyes), and with the depth map render_image makes beside it
            "aa_zoo3d_anaglyph", "aa_c3_random4d_depth"]
# recursive anti-aliasing with a lens: every sample draws its lens point from drand48 (ndt.c:528) -- a stochastic render
AA_LENS_CASES = ["aa_zoo4d_dof"]
# stereo modes, VR / panorama cameras, depth maps (meta: "stereo"; data: "depth" when the case has a depth map)
SAMPLED_CASES = ["ns_c3_random4d", "ns_zoo4d_dof", "al_zoo4d", "al_zoo3d_dof_n3", "ns_zoo4d_sbs", "ns_vr_zoo4d"]   # -n samples > 1 and / or area lights
# ... as an anaglyph, frame-packed, and with a depth map (the LAST sample's hit)
SAMPLED_MODE_CASES = ["ns_zoo3d_anaglyph", "ns_zoo3d_hidef", "ns_c3_random4d_depth"]
VIEW_CASES = ["st_zoo4d_sbs", "st_zoo4d_ou", "st_zoo3d_anaglyph", "st_zoo3d_hidef", "vr_zoo4d", "pano_zoo5d_sbs", "depth_c3_random4d"]
