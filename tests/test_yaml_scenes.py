"""YAML scene files: `ndt -y` writes them, `-s scenes/yaml.so -u file.yaml` reads
them (scene.c:573-2177, scenes/yaml.c).  ndt_amd/host reads them with its own parser.

Fixtures under tests/golden/yaml/ were produced by the compiled reference (tests/golden/make_golden.py):
  <name>.yaml.gz      what the reference's scene_write_yaml wrote (one document per frame)
  <name>.ndtscene.gz  the scene the reference built when it loaded that file back through its own
                      scenes/yaml.so (numbers went through %.16g and atof), flattened
The host's reader must arrive at the same scene, byte for byte: same objects in the same order, same
bounding spheres and kd-tree, same camera.
"""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN
from ndt_amd import load_scene

HOST = os.path.join(ROOT, "ndt_amd", "host")
DRIVER = os.path.join(HOST, "ndt_hip")
YDIR = os.path.join(GOLDEN, "yaml")
CASES = sorted(f[:-5] for f in os.listdir(YDIR) if f.endswith(".json"))
REF_YAML_SO = os.path.join(ROOT, "oracle", "_ref", "scenes", "yaml.so")


@pytest.fixture(scope="module")
def driver():
    subprocess.run(["make", "-C", os.path.join(ROOT, "ndt_amd", "csrc"), "-j", "8"], check=True, capture_output=True)
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    return DRIVER


def _meta(name):
    with open(os.path.join(YDIR, name + ".json")) as f:
        return json.load(f)


def _unpack(name, tmp_path):
    path = str(tmp_path / (name + ".yaml"))
    with gzip.open(os.path.join(YDIR, name + ".yaml.gz"), "rb") as src, open(path, "wb") as dst:
        dst.write(src.read())
    return path


def _fixture_scene_text(name):
    with gzip.open(os.path.join(YDIR, name + ".ndtscene.gz"), "rt") as f:
        return f.read()


def _dump(driver, scene, yaml_path, dims, frame, out):
    cmd = [driver, "-s", scene, "-u", yaml_path, "-d", str(dims), "-f", "%d:%d" % (frame, frame), "--dump-scene", out]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(out))
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        return f.read(), r


@pytest.mark.parametrize("name", CASES)
def test_yaml_loads_to_the_scene_the_reference_builds(driver, tmp_path, name):
    m = _meta(name)
    y = _unpack(name, tmp_path)
    text, _ = _dump(driver, "builtin:yaml", y, m["dims"], m["frame_loaded"], str(tmp_path / "out.ndtscene"))
    assert text == _fixture_scene_text(name)


@pytest.mark.skipif(not os.path.exists(REF_YAML_SO), reason="oracle/_ref not built (make -C oracle ref)")
def test_the_references_own_yaml_scene_plugin_runs_on_this_host(driver, tmp_path):
    """scenes/yaml.so as the reference builds it (WITH_YAML) resolves scene_read_yaml /
    scene_yaml_count_frames from libndt_host.so."""
    name = "y_hypercube3d_2frames"
    m = _meta(name)
    y = _unpack(name, tmp_path)
    text, r = _dump(driver, REF_YAML_SO, y, m["dims"], m["frame_loaded"], str(tmp_path / "out.ndtscene"))
    assert text == _fixture_scene_text(name)


def test_frames_are_documents(driver, tmp_path):
    """scene_yaml_count_frames: one `---` document per animation frame; frame 0 and frame 1 differ."""
    name = "y_hypercube3d_2frames"
    y = _unpack(name, tmp_path)
    a, _ = _dump(driver, "builtin:yaml", y, 3, 0, str(tmp_path / "a.ndtscene"))
    b, _ = _dump(driver, "builtin:yaml", y, 3, 1, str(tmp_path / "b.ndtscene"))
    assert b == _fixture_scene_text(name) and a != b
    # asking for the last frame without naming it: the driver takes the count from the file
    r = subprocess.run([driver, "-s", "builtin:yaml", "-u", y, "-d", "3", "--dump-scene", str(tmp_path / "c.ndtscene")],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert open(tmp_path / "c.ndtscene").read() == b


def test_written_yaml_reads_back_to_the_same_scene(driver, tmp_path):
    """scene_write_yaml -> scene_read_yaml round trip through this host: %.16g in, atof out, twice."""
    prog = tmp_path / "roundtrip"
    code = r'''
#include "ndt_host_api.h"
int main(int argc, char **argv) {
    scene a, b;
    register_objects("objects");
    scene_init(&a, "nameless", atoi(argv[2]));
    scene_read_yaml(&a, argv[1], 0);
    scene_write_yaml(&a, argv[3]);
    scene_init(&b, "nameless", atoi(argv[2]));
    scene_read_yaml(&b, argv[3], 0);
    scene_write_yaml(&b, argv[4]);
    return !(a.num_objects == b.num_objects && a.num_lights == b.num_lights);
}
'''
    cfile = tmp_path / "roundtrip.c"
    cfile.write_text(code)
    r = subprocess.run(["gcc", "-std=c99", "-D_GNU_SOURCE", "-I", os.path.join(HOST, "include"), "-o", str(prog), str(cfile),
                        "-L", HOST, "-lndt_host", "-L", os.path.join(ROOT, "ndt_amd"), "-lndt_hip",
                        "-Wl,-rpath," + HOST, "-Wl,-rpath," + os.path.join(ROOT, "ndt_amd"), "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    y = _unpack("y_zoo5d", tmp_path)
    r = subprocess.run([str(prog), y, "5", str(tmp_path / "w1.yaml"), str(tmp_path / "w2.yaml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    w1, w2 = (tmp_path / "w1.yaml").read_text(), (tmp_path / "w2.yaml").read_text()
    assert w1 == w2 and "objects:" in w1 and "LIGHT_SPOT" in w1
    # and the written file builds the same scene as the reference's file
    text, _ = _dump(driver, "builtin:yaml", str(tmp_path / "w1.yaml"), 5, 0, str(tmp_path / "out.ndtscene"))
    assert text == _fixture_scene_text("y_zoo5d")


@pytest.mark.gpu
def test_yaml_scene_renders_to_the_references_pixels(driver, tmp_path):
    name = "y_random4d"
    m = _meta(name)
    y = _unpack(name, tmp_path)
    raw = str(tmp_path / "fb.f64")
    cmd = [driver, "-s", "builtin:yaml", "-u", y, "-d", str(m["dims"]), "-f", "0:0", "-r", "%dx%d" % (m["width"], m["height"]),
           "-l", str(m["depth"]), "--raw", raw]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    fb = np.fromfile(raw).reshape(m["height"], m["width"], 4)
    ref = np.load(os.path.join(YDIR, name + ".npz"))["fb"]
    assert np.abs(fb - ref).max() < 1e-9
