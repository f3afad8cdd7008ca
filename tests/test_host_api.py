"""The host half of the drop-in: ndt's scene / object / camera C API re-implemented in
ndt_amd/host (libndt_host.so + the ndt_hip driver).

Scene programs are the reference's own, unmodified:
  * built from source against THIS repo's headers (needs /root/reference: build container), and
  * the binaries oracle/_ref/scenes/*.so that were compiled against the REFERENCE's headers
    (struct layouts are ABI; these travel to the GPU box).
Either way the flattened scene our host produces -- bounding spheres from our Nelder-Mead,
our kd-tree build, our camera_aim, our hcube faces -- must equal, byte for byte, the scene the
compiled reference produced (tests/golden/*.ndtscene.gz).
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN, golden

HOST = os.path.join(ROOT, "ndt_amd", "host")
DRIVER = os.path.join(HOST, "ndt_hip")
REF_SRC = "/root/reference/scenes"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "scenes")

# fixture -> (scene program, dims, frame, config)
CASES = {
    "c1_hypercube3d": ("hypercube", 3, 0, None),
    "c1_hypercube3d_f37": ("hypercube", 3, 37, None),
    "c2_balls4d": ("balls", 4, 0, None),
    "c3_random4d": ("random", 4, 0, None),
    "c5_hypercube4d": ("hypercube", 4, 0, None),
    "c5_hypercube5d": ("hypercube", 5, 0, None),
    "c5_hypercube6d": ("hypercube", 6, 0, None),
    "c5_hypercube7d": ("hypercube", 7, 0, None),
    # this repository's own scene program (tests/scenes/parity_zoo.c)
    "zoo4d": ("parity_zoo", 4, 0, None),
    "zoo3d_mirror": ("parity_zoo", 3, 0, "mirror"),
    "zoo5d_f2": ("parity_zoo", 5, 2, None),
    "zoo6d": ("parity_zoo", 6, 0, None),
    # `ndtscene 2` fixtures: the camera2 block (eyes, local axes, fields of view) must match too
    "st_zoo4d_sbs": ("parity_zoo", 4, 0, None),
    "st_zoo3d_anaglyph": ("parity_zoo", 3, 0, None),
    "vr_zoo4d": ("parity_zoo", 4, 0, "vr"),
    "pano_zoo5d_sbs": ("parity_zoo", 5, 0, "pano"),
}
V2 = {"st_zoo4d_sbs", "st_zoo3d_anaglyph", "vr_zoo4d", "pano_zoo5d_sbs"}
OWN_SRC = os.path.join(ROOT, "tests", "scenes")


def _source_of(prog):
    own = os.path.join(OWN_SRC, prog + ".c")
    return own if os.path.exists(own) else os.path.join(REF_SRC, prog + ".c")


@pytest.fixture(scope="module")
def driver():
    subprocess.run(["make", "-C", os.path.join(ROOT, "ndt_amd", "csrc"), "-j", "8"], check=True, capture_output=True)
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    assert os.path.exists(DRIVER)
    return DRIVER


def _fixture_text(name):
    with gzip.open(os.path.join(GOLDEN, name + ".ndtscene.gz"), "rt") as f:
        return f.read()


def _dump(driver, scene_so, dims, frame, out, config=None, v2=False, threads=None):
    cmd = [driver, "-s", scene_so, "-d", str(dims), "-f", "%d:%d" % (frame, frame), "--dump-scene", out]
    if threads:
        cmd += ["-t", str(threads)]
    if config:
        cmd += ["-u", config]
    env = dict(os.environ)
    if v2:
        env["NDT_NDTSCENE_V2"] = "1"       # plain pinhole scenes are written as version 1 unless asked
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(out), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        return f.read()


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_scene_sources_compile_unchanged_and_flatten_identically(driver, tmp_path, name):
    prog, dims, frame, config = CASES[name]
    if not os.path.exists(_source_of(prog)):
        pytest.skip("reference sources only exist in the build container")
    # scenes say #include "../scene.h": lay the tree out so that resolves to our headers
    (tmp_path / "scenes").mkdir()
    for h in os.listdir(os.path.join(HOST, "include")):
        os.symlink(os.path.join(HOST, "include", h), tmp_path / h)
    os.symlink(_source_of(prog), tmp_path / "scenes" / (prog + ".c"))
    so = str(tmp_path / "scenes" / (prog + ".so"))
    r = subprocess.run(["gcc", "-O2", "-std=c99", "-D_GNU_SOURCE", "-fPIC", "-shared", "-Wall", "-o", so,
                        str(tmp_path / "scenes" / (prog + ".c"))], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _dump(driver, so, dims, frame, str(tmp_path / "out.ndtscene"), config, name in V2) == _fixture_text(name)


@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built (make -C oracle ref)")
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_built_scene_binaries_load_unchanged(driver, tmp_path, name):
    prog, dims, frame, config = CASES[name]
    so = os.path.join(REF_BIN, prog + ".so")
    assert _dump(driver, so, dims, frame, str(tmp_path / "out.ndtscene"), config, name in V2) == _fixture_text(name)


@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built (make -C oracle ref)")
@pytest.mark.parametrize("name", ["c5_hypercube6d", "c5_hypercube7d", "c3_random4d"])
def test_sphere_fits_on_several_threads_give_the_same_scene(driver, tmp_path, name):
    """`-t T` fits the lazily fitted bounding spheres on T threads ahead of the flattening loops (ndt_flatten.c): the scene
    that comes out -- spheres, kd-tree, everything -- is the fixture's, whatever T.  (The 6-D / 7-D hypercubes are also what
    exercises the kd-tree build's sorted split scores with inverted boxes: infinite hcylinders among a cluster's children.)"""
    prog, dims, frame, config = CASES[name]
    so = os.path.join(REF_BIN, prog + ".so")
    for threads in (2, 7, 16):
        assert _dump(driver, so, dims, frame, str(tmp_path / ("out%d.ndtscene" % threads)), config, name in V2,
                     threads=threads) == _fixture_text(name)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["c1_hypercube3d", "c2_balls4d", "c3_random4d", "c5_hypercube6d", "zoo4d", "zoo3d_mirror"])
def test_end_to_end_reference_scene_to_pixels(driver, tmp_path, name):
    """The whole drop-in: the reference's scene binary -> our host API -> flatten -> GPU ->
    the framebuffer the compiled reference rendered."""
    prog, dims, frame, config = CASES[name]
    g = golden(name)
    raw = str(tmp_path / "fb.f64")
    cmd = [driver, "-s", os.path.join(REF_BIN, prog + ".so"), "-d", str(dims), "-f", "0",
           "-r", "%dx%d" % (g.width, g.height), "-l", str(g.depth), "--raw", raw]
    if config:
        cmd += ["-u", config]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    fb = np.fromfile(raw).reshape(g.height, g.width, 4)
    assert np.abs(fb - g.data["fb"]).max() < 1e-9
    ppm = [p for p in (tmp_path / "images").rglob("*.ppm")]
    assert len(ppm) == 1


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,flags", [("st_zoo4d_sbs", ["-m", "s"]), ("st_zoo3d_anaglyph", ["-m", "a", "-z"]),
                                        ("vr_zoo4d", ["-z"])])
def test_driver_stereo_and_depth_flags(driver, tmp_path, name, flags):
    """`ndt_hip -m s|o|a` and `-z` (ndt.c:1533-1573, 1726-1729) end to end against the reference's pixels."""
    prog, dims, frame, config = CASES[name]
    g = golden(name)
    raw = str(tmp_path / "fb.f64")
    cmd = [driver, "-s", os.path.join(REF_BIN, prog + ".so"), "-d", str(dims), "-f", "0", "-r", "%dx%d" % (g.width, g.height),
           "-l", str(g.depth), "--raw", raw] + flags
    if config:
        cmd += ["-u", config]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    fb = np.fromfile(raw).reshape(g.height, g.width, 4)
    assert np.abs(fb - g.data["fb"]).max() < 1e-7
    if "-z" in flags:
        dm = np.fromfile(raw + ".depth").reshape(g.height, g.width)
        assert np.abs(dm - g.data["depth"]).max() < 1e-9
        assert len(list((tmp_path / "depth").glob("*.ppm"))) == 1


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built")
def test_frames_in_flight_produce_the_same_images(driver, tmp_path):
    """`ndt_hip -j K`: the scene program runs frame after frame on one thread, K workers (one GPU
    context each) flatten, render and save -- the images must not depend on K."""
    outs = []
    for j in (1, 3):
        d = tmp_path / ("j%d" % j)
        d.mkdir()
        cmd = [driver, "-s", os.path.join(REF_BIN, "hypercube.so"), "-d", "3", "-r", "96x64", "-l", "16", "-f", "30:35", "-j", str(j)]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
        files = sorted(p for p in (d / "images").rglob("*.ppm"))
        assert len(files) == 6
        outs.append([p.read_bytes() for p in files])
    assert outs[0] == outs[1]
    assert len(set(outs[0])) == 6       # an animation: every frame differs


def _read_png_rgba(path):
    """Minimal PNG reader for what ndt_hip --png writes (8-bit RGBA, filter 0 on every row)."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            w, h, bits, colour = struct.unpack(">IIBB", body[:10])
            assert (bits, colour) == (8, 6)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built")
def test_png_at_1080p_is_the_references_image_byte_for_byte(driver, tmp_path):
    """BASELINE config 3 end to end: the reference's scene binary -> this host -> the GPU -> an 8-bit RGBA PNG
    whose every byte is the one the compiled reference's image holds (pixel_d2c of its framebuffer)."""
    g = golden("c3_random4d_1080p")
    cmd = [driver, "-s", os.path.join(REF_BIN, "random.so"), "-d", "4", "-f", "0", "-r", "1920x1080", "-l", str(g.depth), "--png"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    files = list((tmp_path / "images").rglob("*.png"))
    assert len(files) == 1
    assert np.array_equal(_read_png_rgba(str(files[0])), g.data["rgba8"])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(REF_BIN), reason="oracle/_ref not built")
def test_one_frame_over_several_gpu_contexts_from_the_c_driver(driver, tmp_path):
    """`ndt_hip -g N`: ONE frame spread over N GPU contexts by the C host (rows dealt cyclically like the reference's
    MPI_MODE_ROW, ndt.c:812-820; context k on device k mod device count -- on a one-GPU box they share the card), each
    context pushing its rows into the assembled frame.  BASELINE configs[3] (random 4-D, 3840x2160, -l 4) must come out
    as the compiled reference's image, byte for byte, and `-g 1` / `-g 3` must agree on a frame whose height 3 does not divide."""
    g = golden("c4_random4d_4k")
    d = tmp_path / "g8"
    d.mkdir()
    cmd = [driver, "-s", os.path.join(REF_BIN, "random.so"), "-d", "4", "-f", "0", "-r", "4k", "-l", str(g.depth), "--png", "-g", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d))
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    assert "one frame over 8 GPU contexts" in r.stdout
    files = list((d / "images").rglob("*.png"))
    assert len(files) == 1
    got = _read_png_rgba(str(files[0]))
    mism = int((got != g.data["rgba8"]).sum())
    print("4K over 8 contexts: %d of %d bytes differ from the reference's image" % (mism, got.size))
    assert mism == 0
    outs = []
    for n in (1, 3):
        d = tmp_path / ("n%d" % n)
        d.mkdir()
        cmd = [driver, "-s", os.path.join(REF_BIN, "hypercube.so"), "-d", "3", "-f", "0", "-r", "100x64", "-l", "16", "-g", str(n),
               "--raw", str(d / "fb.f64")]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
        outs.append((np.fromfile(str(d / "fb.f64")), [q.read_bytes() for q in sorted((d / "images").rglob("*.ppm"))]))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and len(outs[0][1]) == 1


def _compile_against_host_api(tmp_path, src, out, extra=()):
    """scene programs and object plugins say #include "../scene.h" / "../object.h": lay the tree out so that resolves to
    this repository's host headers"""
    sub = tmp_path / "src"
    sub.mkdir(exist_ok=True)
    for h in os.listdir(os.path.join(HOST, "include")):
        if not os.path.exists(tmp_path / h):
            os.symlink(os.path.join(HOST, "include", h), tmp_path / h)
    dst = sub / os.path.basename(src)
    if not os.path.exists(dst):
        os.symlink(src, dst)
    r = subprocess.run(["gcc", "-O2", "-std=c99", "-D_GNU_SOURCE", "-fPIC", "-shared", "-Wall", *extra, "-o", out, str(dst)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_object_plugins_are_loaded_and_foreign_types_are_refused_by_name(driver, tmp_path):
    """register_objects (object.c:119-153): `-o dir` dlopens every .so of the directory.  A plugin whose type is one of the
    built-in ones stands for the built-in implementation (its scene flattens like any other); a plugin with a type of its
    own loads host-side -- a scene program can allocate it, it gets its bounding sphere and its place in the kd-tree --
    and rendering refuses it by type and file: its intersect() is host code."""
    plug = os.path.join(ROOT, "tests", "plugins")
    objs = tmp_path / "objects"
    objs.mkdir()
    _compile_against_host_api(tmp_path, os.path.join(plug, "blob.c"), str(objs / "blob.so"), ['-DNDT_TEST_PLUGIN_TYPE="blob"'])
    _compile_against_host_api(tmp_path, os.path.join(plug, "blob.c"), str(objs / "my_sphere.so"), ['-DNDT_TEST_PLUGIN_TYPE="sphere"'])
    (objs / "not_a_plugin.so").write_bytes(b"junk")         # the reference prints dlerror() and goes on (object.c:57-60)
    scene_so = str(tmp_path / "plug_scene.so")
    _compile_against_host_api(tmp_path, os.path.join(plug, "plug_scene.c"), scene_so)
    out = str(tmp_path / "out.ndtscene")
    base = [driver, "-s", scene_so, "-d", "4", "-f", "0:0", "-o", str(objs), "--dump-scene", out]
    ok = subprocess.run(base + ["-u", "sphere"], capture_output=True, text=True, cwd=str(tmp_path))
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert "object type 'sphere' from 'my_sphere.so': built in" in ok.stdout
    assert "loaded object type 'blob' from 'blob.so'" in ok.stdout
    assert "sphere 2" in open(out).read() or "sphere" in open(out).read()
    bad = subprocess.run(base + ["-u", "blob"], capture_output=True, text=True, cwd=str(tmp_path))
    assert bad.returncode != 0
    assert "type 'blob' from the plugin 'blob.so'" in (bad.stderr + bad.stdout)
    # without -o the type does not exist at all: the reference's message and exit (object.c:233-236)
    (tmp_path / "elsewhere").mkdir()                        # (the default directory is ./objects, like the reference's)
    none = subprocess.run([driver, "-s", scene_so, "-d", "4", "-f", "0:0", "--dump-scene", out, "-u", "blob"],
                          capture_output=True, text=True, cwd=str(tmp_path / "elsewhere"))
    assert none.returncode != 0 and "Unknown object type 'blob'" in none.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_BIN, "..", "objects")), reason="oracle/_ref not built (make -C oracle ref)")
def test_the_references_own_object_plugins_are_all_built_in(driver, tmp_path):
    """`-o` pointed at the compiled reference's objects/ directory: every plugin there names a built-in type."""
    so = os.path.join(REF_BIN, "random.so")
    out = str(tmp_path / "out.ndtscene")
    r = subprocess.run([driver, "-s", so, "-d", "4", "-f", "0:0", "-o", os.path.abspath(os.path.join(REF_BIN, "..", "objects")),
                        "--dump-scene", out], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "host side only" not in r.stdout
    assert r.stdout.count("built in (device intersector)") >= 9
    assert open(out).read() == _fixture_text("c3_random4d")
