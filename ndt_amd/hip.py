"""ctypes binding of libndt_hip.so (include/ndt_hip.h) -- the product's only compute path.

There is deliberately no fallback: if the shared library is missing or no MI355X is present,
construction raises.  The CPU restatement under oracle/ is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

import numpy as np

from .flat_scene import RenderParams, RenderStats, shard_rows

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NDT_HIP_LIB") or os.path.join(_HERE, "libndt_hip.so")

# every entry point include/ndt_hip.h declares
API_SYMBOLS = [
    "ndt_hip_create", "ndt_hip_destroy", "ndt_hip_upload_scene", "ndt_hip_render_device", "ndt_hip_render",
    "ndt_hip_trace_rays", "ndt_hip_quantize_device", "ndt_hip_shard_rows", "ndt_hip_stream",
    "ndt_hip_synchronize", "ndt_hip_last_error", "ndt_hip_abi_version", "ndt_hip_hcube_hull_box", "ndt_hip_hcube_face_boxes", "ndt_hip_hcube_face_boxes_all", "ndt_hip_hcube_face_tree", "ndt_hip_hcube_face_groups",
    "ndt_hip_render_depth_device", "ndt_hip_render_depth", "ndt_hip_render_rgba8", "ndt_hip_render_multi_device",
    "ndt_hip_render_multi", "ndt_hip_device_count", "ndt_hip_device", "ndt_hip_set_option", "ndt_hip_multi_path_taken",
    "ndt_hip_item_boxes", "ndt_hip_render_rgba8_async", "ndt_hip_render_rgba8_wait",
]

IMAGE_F64, IMAGE_RGBA8 = 0, 1      # enum ndt_image_format
MULTI_NONE, MULTI_LOCAL, MULTI_PEER, MULTI_STAGED = 0, 1, 2, 3     # enum ndt_multi_path


class NdtHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libndt_hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def load_library():
    """Load libndt_hip.so (built in-tree by __graft_entry__.build() / ndt_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `make -C ndt_amd/csrc` (or __graft_entry__.build()). "
            "ndt_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 +
    # libhsa-runtime64.so, and a process that loads /opt/rocm's copy first and torch's second
    # ends up with two HSA runtimes ("No HIP GPUs are available").  Importing torch first makes
    # the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.ndt_hip_last_error.restype = C.c_char_p
    lib.ndt_hip_stream.restype = C.c_void_p
    lib.ndt_hip_stream.argtypes = [C.c_void_p]
    lib.ndt_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.ndt_hip_destroy.argtypes = [C.c_void_p]
    lib.ndt_hip_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_depth_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_trace_rays.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
    lib.ndt_hip_quantize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ndt_hip_synchronize.argtypes = [C.c_void_p]
    lib.ndt_hip_shard_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.ndt_hip_hcube_hull_box.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ndt_hip_hcube_face_boxes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ndt_hip_hcube_face_boxes_all.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    lib.ndt_hip_hcube_face_boxes_all.restype = C.c_int64
    if hasattr(lib, "ndt_hip_hcube_face_groups"):
        lib.ndt_hip_hcube_face_groups.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ndt_hip_hcube_face_groups.restype = C.c_int
    if hasattr(lib, "ndt_hip_hcube_face_tree"):     # (absent from round 3's library, which profiles/ab_libs.sh still loads to compare builds)
        lib.ndt_hip_hcube_face_tree.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ndt_hip_hcube_face_tree.restype = C.c_int64
    lib.ndt_hip_render_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_multi.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_multi_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ndt_hip_device.argtypes = [C.c_void_p]
    lib.ndt_hip_multi_path_taken.argtypes = [C.c_void_p]
    lib.ndt_hip_item_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_rgba8_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndt_hip_render_rgba8_wait.argtypes = [C.c_void_p]
    lib.ndt_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    _lib = lib
    return lib


def hcube_hull_box(fs, obj):
    """The hull box libndt_hip derives for hcube `obj` of FlatScene `fs` (host only, no GPU):
    (axes[N,N], centre[N], half[N]) or None when the hcube gets no box."""
    import numpy as np
    lib = load_library()
    n = fs.dims
    rows = np.zeros((n, n + 2), dtype=np.float64)
    rc = lib.ndt_hip_hcube_hull_box(fs.byref(), int(obj), rows.ctypes.data)
    if rc < 0:
        raise NdtHipError(rc, (lib.ndt_hip_last_error() or b"").decode())
    if rc == 0:
        return None
    return rows[:, :n].copy(), rows[:, n].copy(), rows[:, n + 1].copy()


def item_boxes(fs):
    """ndt_hip_item_boxes: (frame [d, d], rows [n_items, d, 2], has [n_items] bool) or None when the scene has no boxed item."""
    lib = load_library()
    d, n = fs.dims, fs.struct.n_items
    frame = np.zeros((d, d))
    rows = np.zeros((max(n, 1), d, 2))
    has = np.zeros(max(n, 1), dtype=np.uint8)
    rc = lib.ndt_hip_item_boxes(fs.byref(), frame.ctypes.data, rows.ctypes.data, has.ctypes.data)
    if rc < 0:
        raise NdtHipError(rc, (lib.ndt_hip_last_error() or b"").decode())
    if rc == 0:
        return None
    return frame, rows[:n], has[:n].astype(bool)


def hcube_face_boxes(fs, obj):
    """The boxes of the single faces of hcube `obj` in its hull box's frame (host only, no GPU), for an hcube of any number of
    faces: (centre[F,N], half[F,N], possible[F] bool) or None when the hcube gets no face boxes."""
    import numpy as np
    lib = load_library()
    n = fs.dims
    count = lib.ndt_hip_hcube_face_boxes_all(fs.byref(), int(obj), 0, None, None)
    if count < 0:
        raise NdtHipError(int(count), (lib.ndt_hip_last_error() or b"").decode())
    if count == 0:
        return None
    rows = np.zeros((count, n, 2), dtype=np.float64)
    possible = np.zeros(count, dtype=np.uint8)
    rc = lib.ndt_hip_hcube_face_boxes_all(fs.byref(), int(obj), count, rows.ctypes.data, possible.ctypes.data)
    assert rc == count
    return rows[:, :, 0].copy(), rows[:, :, 1].copy(), possible.astype(bool)


def hcube_face_tree(fs, obj):
    """The hierarchy over the face boxes of hcube `obj` (host only, no GPU): a list, level j = 1 .. top, of
    (centre[K_j, N], half[K_j, N]) for the aligned runs of 2^j faces; None when the hcube gets no face boxes."""
    import numpy as np
    lib = load_library()
    n = fs.dims
    top = C.c_int32(0)
    off = (C.c_int32 * 32)()
    count = lib.ndt_hip_hcube_face_tree(fs.byref(), int(obj), 0, None, off, C.byref(top))
    if count < 0:
        raise NdtHipError(int(count), (lib.ndt_hip_last_error() or b"").decode())
    if count == 0:
        return None
    rows = np.zeros((count, n, 2), dtype=np.float64)
    rc = lib.ndt_hip_hcube_face_tree(fs.byref(), int(obj), count, rows.ctypes.data, off, C.byref(top))
    assert rc == count
    levels = []
    for j in range(1, top.value + 1):
        end = off[j + 1] if j < top.value else count
        levels.append((rows[off[j]:end, :, 0].copy(), rows[off[j]:end, :, 1].copy()))
    return levels


def hcube_face_groups(fs, obj):
    """The index of hcube `obj`'s faces by the hull axes their boxes are thin on (host only, no GPU):
    (clusters[N, 2, 2] = per axis and side {centre, half}, table[2^N, 2] = {start, count} per subset of the axes into members,
    face_set[n_faces], members); None when the hcube gets no face boxes."""
    import numpy as np
    lib = load_library()
    n = fs.dims
    nf = fs.objects[int(obj)]["n_obj"]
    clusters = np.zeros((n, 2, 2), dtype=np.float64)
    table = np.zeros((1 << n, 2), dtype=np.int32)
    face_set = np.zeros(max(nf, 1), dtype=np.int32)
    members = np.zeros(max(nf, 1), dtype=np.int32)
    rc = lib.ndt_hip_hcube_face_groups(fs.byref(), int(obj), clusters.ctypes.data, table.ctypes.data, face_set.ctypes.data, members.ctypes.data)
    if rc < 0:
        raise NdtHipError(int(rc), (lib.ndt_hip_last_error() or b"").decode())
    if rc == 0:
        return None
    return clusters, table, face_set[:rc], members[:int((face_set[:rc] >= 0).sum())]


class NdtHip:
    """One rendering context = one MI355X + one HIP stream (ndt_hip_create)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        self._check(self.lib.ndt_hip_create(int(device), C.byref(self.ctx)))
        self.device = int(device)
        self.scene = None

    def _check(self, rc):
        if rc != 0:
            raise NdtHipError(rc, (self.lib.ndt_hip_last_error() or b"").decode())

    def close(self):
        if self.ctx:
            self.lib.ndt_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self.lib.ndt_hip_stream(self.ctx)

    def synchronize(self):
        self._check(self.lib.ndt_hip_synchronize(self.ctx))

    def set_option(self, name, value):
        """ndt_hip_set_option: "pipeline" (0 auto, 1 levels, 2 stream, 3 hybrid), "hull_box", "face_box", "debug_levels", ..."""
        self._check(self.lib.ndt_hip_set_option(self.ctx, name.encode(), int(value)))

    def multi_path_taken(self):
        """ndt_hip_multi_path_taken: how this context's rows reached the frame in the last render_multi (MULTI_*)."""
        return int(self.lib.ndt_hip_multi_path_taken(self.ctx))

    def upload_scene(self, fs):
        self._check(self.lib.ndt_hip_upload_scene(self.ctx, fs.byref()))
        self.scene = fs          # keep the arrays alive; also gives dims

    def params(self, width, height, depth, row_begin=0, row_step=1, specular=1, profile=0, aa=None, stereo=0, samples=1):
        p = RenderParams(width, height, depth, int(samples), row_begin, row_step, specular, profile)
        p.stereo = int(stereo)      # ndt_stereo_mode: 0 mono, 1 side by side, 2 over/under, 3 anaglyph
        if aa is not None:
            # Whitted's recursive anti-aliasing, `-a diff,depth` (ndt.c:655-733)
            p.recursive_aa, p.aa_diff, p.aa_depth = 1, int(aa[0]), int(aa[1])
        return p

    def render(self, width, height, depth, row_begin=0, row_step=1, specular=1, profile=0, aa=None, stereo=0,
               depth_map=False, samples=1):
        """render_image for a row shard; returns ((rows, width, 4) float64 host array, RenderStats).
        aa = (aa_diff, aa_depth) switches recursive anti-aliasing on; stereo = ndt_stereo_mode;
        depth_map=True returns (rgba, (rows, width) depth map, stats)."""
        p = self.params(width, height, depth, row_begin, row_step, specular, profile, aa, stereo, samples)
        if depth_map:
            rows = shard_rows(height, row_begin, row_step)
            out = np.zeros((rows, width, 4), dtype=np.float64)
            dm = np.zeros((rows, width), dtype=np.float64)
            st = RenderStats()
            self._check(self.lib.ndt_hip_render_depth(self.ctx, C.byref(p), out.ctypes.data_as(C.c_void_p),
                                                      dm.ctypes.data_as(C.c_void_p), C.byref(st)))
            return out, dm, st
        rows = shard_rows(height, row_begin, row_step)
        out = np.zeros((rows, width, 4), dtype=np.float64)
        st = RenderStats()
        self._check(self.lib.ndt_hip_render(self.ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st

    def render_device(self, d_rgba_ptr, width, height, depth, row_begin=0, row_step=1, specular=1, profile=0, aa=None, stereo=0,
                      samples=1):
        """Same, output left in HBM at raw device pointer `d_rgba_ptr` (rows*width*4 doubles)."""
        p = self.params(width, height, depth, row_begin, row_step, specular, profile, aa, stereo, samples)
        st = RenderStats()
        self._check(self.lib.ndt_hip_render_device(self.ctx, C.byref(p), C.c_void_p(d_rgba_ptr), C.byref(st)))
        return st

    def render_rgba8(self, width, height, depth, **kw):
        """render_image + the reference's save-time quantisation on the device: (rows, width, 4) uint8 host array, stats."""
        p = self.params(width, height, depth, **kw)
        rows = shard_rows(height, p.row_begin, p.row_step)
        out = np.zeros((rows, width, 4), dtype=np.uint8)
        st = RenderStats()
        self._check(self.lib.ndt_hip_render_rgba8(self.ctx, C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, st

    def render_rgba8_async(self, host_ptr, width, height, depth, **kw):
        """ndt_hip_render_rgba8_async: the frame's bytes travel to pinned host memory at `host_ptr` behind the next call's
        rendering; read them after render_rgba8_wait()."""
        p = self.params(width, height, depth, **kw)
        st = RenderStats()
        self._check(self.lib.ndt_hip_render_rgba8_async(self.ctx, C.byref(p), C.c_void_p(host_ptr), C.byref(st)))
        return st

    def render_rgba8_wait(self):
        self._check(self.lib.ndt_hip_render_rgba8_wait(self.ctx))

    def quantize_device(self, d_rgba_ptr, d_rgba8_ptr, n_pixels):
        self._check(self.lib.ndt_hip_quantize_device(self.ctx, C.c_void_p(d_rgba_ptr), C.c_void_p(d_rgba8_ptr),
                                                     int(n_pixels)))

    def trace_rays(self, rays):
        """Batch of trace_kd queries; rays: (n, 2*dims+1) = o, v, dist_limit per row."""
        d = self.scene.dims
        n = rays.shape[0]
        o = np.ascontiguousarray(rays[:, :d], dtype=np.float64)
        v = np.ascontiguousarray(rays[:, d:2 * d], dtype=np.float64)
        lim = np.ascontiguousarray(rays[:, 2 * d], dtype=np.float64)
        obj = np.zeros(n, dtype=np.int32)
        hit = np.zeros((n, d))
        nrm = np.zeros((n, d))
        self._check(self.lib.ndt_hip_trace_rays(self.ctx, n, o.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                                lim.ctypes.data_as(C.c_void_p), obj.ctypes.data_as(C.c_void_p),
                                                hit.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p)))
        return obj, hit, nrm


def render_multi(contexts, width, height, depth, fmt=IMAGE_F64, d_out_ptr=None, **kw):
    """ONE frame over several NdtHip contexts (one per GPU, or several on one GPU) from this one thread:
    ndt_hip_render_multi.  Rows are dealt cyclically like the reference's MPI_MODE_ROW (ndt.c:812-820); every context
    must hold the same uploaded scene.  Returns (image, stats): float64 (rows, width, 4) or uint8 for IMAGE_RGBA8;
    with d_out_ptr (memory of contexts[0]'s device) the image stays there and only the stats come back."""
    first = contexts[0]
    p = first.params(width, height, depth, **kw)
    arr = (C.c_void_p * len(contexts))(*[c.ctx for c in contexts])
    st = RenderStats()
    if d_out_ptr is not None:
        first._check(first.lib.ndt_hip_render_multi_device(arr, len(contexts), C.byref(p), int(fmt), C.c_void_p(d_out_ptr), C.byref(st)))
        return None, st
    rows = shard_rows(height, p.row_begin, p.row_step)
    out = np.zeros((rows, width, 4), dtype=np.uint8 if fmt == IMAGE_RGBA8 else np.float64)
    first._check(first.lib.ndt_hip_render_multi(arr, len(contexts), C.byref(p), int(fmt), out.ctypes.data_as(C.c_void_p), C.byref(st)))
    return out, st
