"""The modes beyond the deterministic frame at full size on the device, with the compiled reference
(oracle/_ref, same box) beside them on a bounded sample: recursive anti-aliasing -a 20,4 and -n 4 jittered samples
(the benchmark scene has no stereo eyes; the stereo / VR / depth-map modes cost what their passes cost).  usage: python profiles/modes_probe.py [--no-cpu]"""
import os
import re
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from ndt_amd import load_scene
from ndt_amd.hip import NdtHip

W, H, DEPTH = 1920, 1080, 4
cpu = "--no-cpu" not in sys.argv
fs = load_scene(os.path.join(ROOT, "tests", "golden", "c3_random4d.ndtscene.gz"))
g = NdtHip(0)
g.upload_scene(fs)


def device(label, w, h, **kw):
    buf = torch.empty((2 * h + 64, 2 * w, 4), dtype=torch.float64, device="cuda")      # room for the packed stereo images
    g.render_device(buf.data_ptr(), w, h, DEPTH, **kw)
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(buf.data_ptr(), w, h, DEPTH, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    d = st.as_dict()
    rays = d["rays_primary"] + d["rays_secondary"] + d["rays_shadow"]
    print("device  %-34s %4dx%-4d %9.2f ms  %6.1f M rays traced (%7.1f M in the reference's counting)  %5.2f Gray/s" % (
        label, w, h, ms, rays / 1e6, d["rays_ref_equiv"] / 1e6, rays / ms / 1e6), flush=True)
    return ms, d["rays_ref_equiv"]


def reference(label, w, h, threads, extra, full_calls):
    shim = os.path.join(ROOT, "oracle", "_ref", "ndt_ref_shim")
    if not cpu or not os.path.exists(shim):
        return
    cmd = [shim, "--objects", os.path.join(ROOT, "oracle", "_ref", "objects"),
           "--scene", os.path.join(ROOT, "oracle", "_ref", "scenes", "random.so"), "--dims", "4", "--res", "%dx%d" % (w, h),
           "--threads", str(threads), "--depth", str(DEPTH), "--tmp", "/tmp"] + extra
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=900).stdout
    sec = float(re.search(r"ref_shim: render_s ([0-9.]+)", out).group(1))
    rays = int(re.search(r"rays_total (\d+)", out).group(1))
    print("ref CPU %-34s %4dx%-4d %9.2f s   %7.1f M trace_kd calls, %d pthreads: %.2f Mray/s; the 1920x1080 frame's %.1f M calls at that rate: %.0f s" % (
        label, w, h, sec, rays / 1e6, threads, rays / sec / 1e6, full_calls / 1e6, full_calls / (rays / sec)), flush=True)


_, calls = device("plain frame", W, H)
reference("plain frame", W // 2, H // 2, 16, [], calls)
_, calls = device("-a 20,4 (recursive AA)", W, H, aa=(20, 4))
reference("-a 20,4 (recursive AA)", W // 4, H // 4, 16, ["--aa", "20,4"], calls)
_, calls = device("-n 4 (jittered samples)", W, H, samples=4)
reference("-n 4 (jittered samples)", W // 8, H // 8, 1, ["--samples", "4"], calls)
