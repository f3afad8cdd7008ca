"""ctypes mirror of include/ndt_hip.h and the `ndtscene` text format.

`ndtscene` is the on-disk form of an `ndt_flat_scene`: the reference's `scene` / `object` /
`kd_tree_t` graph (scene.h:51-62, object.h:23-74, kd-tree.h:52-71) flattened to index-linked
arrays, one keyword per line, every double as a C99 hex float so values survive exactly (the
reference's own YAML dump prints %.16g, scene.c:682, which does not round-trip).  Files under
tests/golden/ were written by oracle/ref_shim.c from the compiled reference.
"""
import ctypes as C
import gzip
import numpy as np

ABI_VERSION = 3
MIN_DIMS, MAX_DIMS = 3, 12

OBJ_TYPES = ["sphere", "hplane", "hdisk", "cylinder", "hcylinder", "orthotope", "hcube", "hfacet", "facet"]
OBJ_TYPE_ID = {name: i for i, name in enumerate(OBJ_TYPES)}
LIGHT_AMBIENT, LIGHT_POINT, LIGHT_DIRECTIONAL, LIGHT_SPOT, LIGHT_DISK, LIGHT_RECT = range(6)

NDT_OK, NDT_E_INVALID, NDT_E_UNSUPPORTED, NDT_E_DEVICE, NDT_E_NOMEM, NDT_E_STATE = 0, -1, -2, -3, -4, -5


class FlatLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos_off", C.c_int32), ("dir_off", C.c_int32), ("area_off", C.c_int32),
                ("red", C.c_double), ("green", C.c_double), ("blue", C.c_double), ("angle", C.c_double),
                ("radius", C.c_double)]


class FlatObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("transparent", C.c_int32), ("parent", C.c_int32),
                ("n_pos", C.c_int32), ("pos_off", C.c_int32),
                ("n_dir", C.c_int32), ("dir_off", C.c_int32),
                ("n_size", C.c_int32), ("size_off", C.c_int32),
                ("n_flag", C.c_int32), ("flag_off", C.c_int32),
                ("n_obj", C.c_int32), ("obj_off", C.c_int32),
                ("bounds_center_off", C.c_int32),
                ("bounds_radius", C.c_double),
                ("red", C.c_double), ("green", C.c_double), ("blue", C.c_double),
                ("red_r", C.c_double), ("green_r", C.c_double), ("blue_r", C.c_double),
                ("refract_index", C.c_double)]


class FlatKdNode(C.Structure):
    _fields_ = [("dim", C.c_int32), ("num", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
                ("first", C.c_int32), ("_pad", C.c_int32), ("boundary", C.c_double)]


class FlatSceneStruct(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dims", C.c_int32),
                ("vecs", C.POINTER(C.c_double)), ("n_vecs", C.c_int64),
                ("sizes", C.POINTER(C.c_double)), ("n_sizes", C.c_int64),
                ("flags", C.POINTER(C.c_int32)), ("n_flags", C.c_int64),
                ("obj_refs", C.POINTER(C.c_int32)), ("n_obj_refs", C.c_int64),
                ("cam_type", C.c_int32),
                ("cam_pos_off", C.c_int32), ("cam_img_orig_off", C.c_int32),
                ("cam_dir_x_off", C.c_int32), ("cam_dir_y_off", C.c_int32),
                ("cam_focal_distance", C.c_double),
                ("ambient", C.c_double * 3), ("background", C.c_double * 4),
                ("lights", C.POINTER(FlatLight)), ("n_lights", C.c_int32),
                ("objects", C.POINTER(FlatObject)), ("n_objects", C.c_int32), ("n_items", C.c_int32),
                ("kd_nodes", C.POINTER(FlatKdNode)), ("n_kd_nodes", C.c_int32),
                ("leaf_refs", C.POINTER(C.c_int32)), ("n_leaf_refs", C.c_int32),
                ("inf_refs", C.POINTER(C.c_int32)), ("n_inf", C.c_int32),
                ("bb_lower_off", C.c_int32), ("bb_upper_off", C.c_int32),
                ("cam_aperture_radius", C.c_double), ("cam_h_fov", C.c_double), ("cam_v_fov", C.c_double),
                ("cam_left_eye_off", C.c_int32), ("cam_right_eye_off", C.c_int32),
                ("cam_local_x_off", C.c_int32), ("cam_local_y_off", C.c_int32), ("cam_local_z_off", C.c_int32),
                ("_pad2", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_optic_depth", C.c_int32),
                ("samples", C.c_int32), ("row_begin", C.c_int32), ("row_step", C.c_int32),
                ("specular", C.c_int32), ("profile", C.c_int32),
                ("recursive_aa", C.c_int32), ("aa_diff", C.c_int32), ("aa_depth", C.c_int32),
                ("stereo", C.c_int32), ("reserved", C.c_int32 * 4)]


class RenderStats(C.Structure):
    _fields_ = [("rays_primary", C.c_int64), ("rays_secondary", C.c_int64), ("rays_shadow", C.c_int64),
                ("rays_ref_equiv", C.c_int64), ("levels", C.c_int32), ("trace_launches", C.c_int32),
                ("trace_ms", C.c_double), ("frame_ms", C.c_double), ("node_capacity", C.c_int64),
                ("pixels_resampled", C.c_int64), ("aa_samples", C.c_int64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def shard_rows(height, row_begin, row_step):
    """Rows a (row_begin,row_step) shard of a `height`-row image holds (ndt.c:812-820)."""
    if row_begin >= height:
        return 0
    return (height - row_begin + row_step - 1) // row_step


def _hx(tok):
    return float.fromhex(tok)


class FlatScene:
    """A flat scene held in numpy arrays, plus the ctypes view handed across the C ABI.

    Keep the FlatScene object alive for as long as the struct is in use: the struct only
    borrows pointers into the arrays.
    """

    def __init__(self):
        self.name = ""
        self.dims = 0
        self.cam_type = 0
        self.cam_focal_distance = 100.0
        self.cam_aperture_radius = 0.0
        self.cam_h_fov = 0.0
        self.cam_v_fov = 0.0
        self.ambient = [0.0, 0.0, 0.0]
        self.background = [0.0, 0.0, 0.0, 1.0]
        self._vecs = []          # list of float
        self._sizes = []
        self._flags = []
        self._obj_refs = []
        self.lights = []         # dicts
        self.objects = []        # dicts
        self.n_items = 0
        self.kd_nodes = []       # dicts
        self.leaf_refs = []
        self.inf_refs = []
        self.cam = {}
        self.bb = {}
        self._struct = None

    # ---- building ----
    def add_vec(self, values):
        if len(values) != self.dims:
            raise ValueError("vector of %d components in a %d-D scene" % (len(values), self.dims))
        off = len(self._vecs)
        self._vecs.extend(float(x) for x in values)
        return off

    def vec(self, off):
        return np.array(self._vecs[off:off + self.dims], dtype=np.float64)

    @property
    def n_objects(self):
        return len(self.objects)

    def finalize(self):
        d = self.dims
        self.vecs = np.ascontiguousarray(np.array(self._vecs if self._vecs else [0.0], dtype=np.float64))
        self.sizes = np.ascontiguousarray(np.array(self._sizes if self._sizes else [0.0], dtype=np.float64))
        self.flags = np.ascontiguousarray(np.array(self._flags if self._flags else [0], dtype=np.int32))
        self.obj_refs = np.ascontiguousarray(np.array(self._obj_refs if self._obj_refs else [0], dtype=np.int32))
        self.leaf_refs_a = np.ascontiguousarray(np.array(self.leaf_refs if self.leaf_refs else [0], dtype=np.int32))
        self.inf_refs_a = np.ascontiguousarray(np.array(self.inf_refs if self.inf_refs else [0], dtype=np.int32))
        self.lights_a = (FlatLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            s = self.lights_a[i]
            s.type, s.pos_off, s.dir_off = l["type"], l["pos_off"], l["dir_off"]
            s.red, s.green, s.blue, s.angle = l["red"], l["green"], l["blue"], l["angle"]
            s.area_off, s.radius = l.get("area_off", -1), l.get("radius", 0.0)
        self.objects_a = (FlatObject * max(1, len(self.objects)))()
        for i, o in enumerate(self.objects):
            s = self.objects_a[i]
            for key in ("type", "transparent", "parent", "n_pos", "pos_off", "n_dir", "dir_off", "n_size",
                        "size_off", "n_flag", "flag_off", "n_obj", "obj_off", "bounds_center_off",
                        "bounds_radius", "red", "green", "blue", "red_r", "green_r", "blue_r", "refract_index"):
                setattr(s, key, o[key])
        self.kd_a = (FlatKdNode * max(1, len(self.kd_nodes)))()
        for i, k in enumerate(self.kd_nodes):
            s = self.kd_a[i]
            s.dim, s.num, s.left, s.right, s.first, s.boundary = (k["dim"], k["num"], k["left"], k["right"],
                                                                   k["first"], k["boundary"])
        st = FlatSceneStruct()
        st.abi_version = ABI_VERSION
        st.dims = d
        st.vecs = self.vecs.ctypes.data_as(C.POINTER(C.c_double)); st.n_vecs = len(self._vecs)
        st.sizes = self.sizes.ctypes.data_as(C.POINTER(C.c_double)); st.n_sizes = len(self._sizes)
        st.flags = self.flags.ctypes.data_as(C.POINTER(C.c_int32)); st.n_flags = len(self._flags)
        st.obj_refs = self.obj_refs.ctypes.data_as(C.POINTER(C.c_int32)); st.n_obj_refs = len(self._obj_refs)
        st.cam_type = self.cam_type
        st.cam_pos_off = self.cam["pos"]; st.cam_img_orig_off = self.cam["img_orig"]
        st.cam_dir_x_off = self.cam["dir_x"]; st.cam_dir_y_off = self.cam["dir_y"]
        st.cam_focal_distance = self.cam_focal_distance
        for i in range(3):
            st.ambient[i] = self.ambient[i]
        for i in range(4):
            st.background[i] = self.background[i]
        st.lights = C.cast(self.lights_a, C.POINTER(FlatLight)); st.n_lights = len(self.lights)
        st.objects = C.cast(self.objects_a, C.POINTER(FlatObject)); st.n_objects = len(self.objects)
        st.n_items = self.n_items
        st.kd_nodes = C.cast(self.kd_a, C.POINTER(FlatKdNode)); st.n_kd_nodes = len(self.kd_nodes)
        st.leaf_refs = self.leaf_refs_a.ctypes.data_as(C.POINTER(C.c_int32)); st.n_leaf_refs = len(self.leaf_refs)
        st.inf_refs = self.inf_refs_a.ctypes.data_as(C.POINTER(C.c_int32)); st.n_inf = len(self.inf_refs)
        st.bb_lower_off = self.bb["lower"]; st.bb_upper_off = self.bb["upper"]
        st.cam_aperture_radius = self.cam_aperture_radius
        st.cam_h_fov = self.cam_h_fov; st.cam_v_fov = self.cam_v_fov
        st.cam_left_eye_off = self.cam.get("left_eye", -1); st.cam_right_eye_off = self.cam.get("right_eye", -1)
        st.cam_local_x_off = self.cam.get("local_x", -1); st.cam_local_y_off = self.cam.get("local_y", -1)
        st.cam_local_z_off = self.cam.get("local_z", -1)
        self._struct = st
        return self

    @property
    def struct(self):
        if self._struct is None:
            self.finalize()
        return self._struct

    def byref(self):
        return C.byref(self.struct)

    def type_histogram(self):
        h = {}
        for o in self.objects:
            name = OBJ_TYPES[o["type"]] + ("" if o["parent"] < 0 else "(nested)")
            h[name] = h.get(name, 0) + 1
        return h


def _open_text(path):
    path = str(path)
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def load_scene(path):
    """Parse an `ndtscene 1` file (plain or .gz) into a finalized FlatScene."""
    fs = FlatScene()
    with _open_text(path) as f:
        lines = [ln.split() for ln in f if ln.strip()]
    it = iter(lines)

    def need(key):
        tok = next(it)
        if tok[0] != key:
            raise ValueError("%s: expected '%s', got '%s'" % (path, key, tok[0]))
        return tok

    tok = need("ndtscene")
    version = tok[1]
    if version not in ("1", "2"):
        raise ValueError("unsupported ndtscene version " + tok[1])
    tok = need("name"); fs.name = " ".join(tok[1:])
    fs.dims = int(need("dims")[1])
    tok = need("camera"); fs.cam_type = int(tok[2]); fs.cam_focal_distance = _hx(tok[4])
    fs.cam["pos"] = fs.add_vec([_hx(t) for t in need("cam_pos")[1:]])
    fs.cam["img_orig"] = fs.add_vec([_hx(t) for t in need("cam_img_orig")[1:]])
    fs.cam["dir_x"] = fs.add_vec([_hx(t) for t in need("cam_dir_x")[1:]])
    fs.cam["dir_y"] = fs.add_vec([_hx(t) for t in need("cam_dir_y")[1:]])
    if version == "2":
        # the rest of the camera: aperture, fields of view, eyes and local axes (camera.h:34-76)
        tok = need("camera2")
        fs.cam_aperture_radius = _hx(tok[2]); fs.cam_h_fov = _hx(tok[4]); fs.cam_v_fov = _hx(tok[6])
        for key in ("left_eye", "right_eye", "local_x", "local_y", "local_z"):
            fs.cam[key] = fs.add_vec([_hx(t) for t in need("cam_" + key)[1:]])
    fs.ambient = [_hx(t) for t in need("ambient")[1:4]]
    fs.background = [_hx(t) for t in need("background")[1:5]]
    n_lights = int(need("lights")[1])
    for _ in range(n_lights):
        tok = need("light")
        l = {"type": int(tok[3]), "red": _hx(tok[5]), "green": _hx(tok[6]), "blue": _hx(tok[7]),
             "angle": _hx(tok[9])}
        has_pos, has_dir = int(tok[11]), int(tok[13])
        pos = [_hx(t) for t in need("lpos")[1:]]
        dr = [_hx(t) for t in need("ldir")[1:]]
        l["pos_off"] = fs.add_vec(pos) if has_pos else -1
        l["dir_off"] = fs.add_vec(dr) if has_dir else -1
        l["area_off"], l["radius"] = -1, 0.0
        if len(tok) > 17 and int(tok[17]):
            # area light (ndtscene 2 files written since ABI 3): radius, then the prepared basis u1, v1
            l["radius"] = _hx(tok[15])
            l["area_off"] = fs.add_vec([_hx(t) for t in need("lu1")[1:]])
            fs.add_vec([_hx(t) for t in need("lv1")[1:]])
        fs.lights.append(l)
    tok = need("objects")
    n_objects, fs.n_items = int(tok[1]), int(tok[3])
    for i in range(n_objects):
        tok = need("object")
        if int(tok[1]) != i:
            raise ValueError("object numbering")
        o = {"type": OBJ_TYPE_ID[tok[3]], "parent": int(tok[5]), "transparent": int(tok[7]),
             "n_pos": int(tok[9]), "n_dir": int(tok[11]), "n_size": int(tok[13]), "n_flag": int(tok[15]),
             "n_obj": int(tok[17])}
        m = [_hx(t) for t in need("material")[1:8]]
        o["red"], o["green"], o["blue"], o["red_r"], o["green_r"], o["blue_r"], o["refract_index"] = m
        tok = need("bounds")
        o["bounds_radius"] = _hx(tok[1])
        o["bounds_center_off"] = fs.add_vec([_hx(t) for t in tok[2:]])
        o["pos_off"] = len(fs._vecs)
        for _ in range(o["n_pos"]):
            fs.add_vec([_hx(t) for t in need("pos")[1:]])
        o["dir_off"] = len(fs._vecs)
        for _ in range(o["n_dir"]):
            fs.add_vec([_hx(t) for t in need("dir")[1:]])
        tok = need("sizes"); o["size_off"] = len(fs._sizes); fs._sizes.extend(_hx(t) for t in tok[1:])
        tok = need("flags"); o["flag_off"] = len(fs._flags); fs._flags.extend(int(t) for t in tok[1:])
        tok = need("children"); o["obj_off"] = len(fs._obj_refs); fs._obj_refs.extend(int(t) for t in tok[1:])
        if len(tok) - 1 != o["n_obj"]:
            raise ValueError("children count")
        fs.objects.append(o)
    tok = need("kdtree")
    n_nodes = int(tok[2])
    for i in range(n_nodes):
        tok = need("kdnode")
        if int(tok[1]) != i:
            raise ValueError("kd node numbering")
        num = int(tok[11])
        ids = [int(t) for t in tok[13:13 + num]]
        node = {"dim": int(tok[3]), "boundary": _hx(tok[5]), "left": int(tok[7]), "right": int(tok[9]),
                "num": num, "first": len(fs.leaf_refs)}
        fs.leaf_refs.extend(ids)
        fs.kd_nodes.append(node)
    tok = need("inf")
    fs.inf_refs = [int(t) for t in tok[3:3 + int(tok[1])]]
    fs.bb["lower"] = fs.add_vec([_hx(t) for t in need("bb_lower")[1:]])
    fs.bb["upper"] = fs.add_vec([_hx(t) for t in need("bb_upper")[1:]])
    need("end")
    return fs.finalize()
