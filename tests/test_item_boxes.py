"""Item boxes (ndt_blob.hip:scene_item_boxes, ndt_hip_item_boxes): scenes of more than 256 items -- the 6-D .. 8-D hypercubes,
thousands of orthotopes behind bounding spheres as wide as their diagonals -- carry ONE orthonormal frame and, per orthotope,
the box in that frame of every point its intersect() can return.  A ray that misses an item's box skips the item's
bounding-sphere gate and its intersect() (objects/orthotope.c:150-302): like the hcube hull box (test_hull_box.py) that is
only legitimate if the reference can never return a point outside the box.  Checked here
  * on the CPU: every hit the oracle reports on an orthotope, for rays aimed at the orthotopes, lies inside its box with the
    slack the derivation promises (margin 0.02 against a reach of 0.01485); the frame is orthonormal; for a rotated cube it
    is the cube's own frame;
  * on the GPU: the same rays -- and the known answers of the 6-D / 7-D scenes -- come out bit-identical with the boxes on,
    with the boxes off ("item_boxes" 0) and from the oracle; also for a scene whose objects were turned out of the
    coordinate axes (the frame is then not the world's), and whether or not the lanes of a wavefront scan their leaf
    together ("leaf_scan").
"""
import os

import numpy as np
import pytest

from conftest import golden, GOLDEN

ORTHOTOPE = 5
SLACK = 0.02 - 0.01485
SCENES = ["c5_hypercube6d", "c1_hypercube3d_f37", "c5_hypercube4d", "zoo6d", "c3_random4d"]


def orthotopes(fs):
    n_items = fs.struct.n_items
    return [i for i, o in enumerate(fs.objects[:n_items]) if o["type"] == ORTHOTOPE]


def aimed_rays(fs, ids, seed, per_target=3, limit=400):
    """Rays aimed at where orthotopes can be hit: around their corner, over their span, and grazing their edges."""
    rng = np.random.default_rng(seed)
    n = fs.dims
    vecs = np.asarray(fs.vecs, dtype=np.float64).ravel()
    flags = np.asarray(fs.flags)
    if len(ids) > limit:
        ids = list(rng.choice(ids, limit, replace=False))
    targets = []
    for i in ids:
        f = fs.objects[i]
        m = int(flags[f["flag_off"]])
        pos = vecs[f["pos_off"]:f["pos_off"] + n]
        dirs = vecs[f["dir_off"]:f["dir_off"] + m * n].reshape(m, n)
        for _ in range(per_target):
            targets.append(pos + rng.normal(0, 0.006, n))
            targets.append(pos + rng.uniform(0, 1, m) @ dirs + rng.normal(0, 0.004, n))
            edge = rng.integers(0, 2, m).astype(float)
            edge[rng.integers(0, m)] = rng.uniform(0, 1)
            targets.append(pos + edge @ dirs + rng.normal(0, 0.008, n))
    targets = np.array(targets)
    d = rng.normal(0, 1, targets.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.uniform(2.0, 12.0, (len(targets), 1))
    rays = np.zeros((len(targets), 2 * n + 1))
    rays[:, :n] = targets - d * dist
    rays[:, n:2 * n] = d
    rays[:, 2 * n] = -1.0
    rays[::3, 2 * n] = dist[::3, 0] * 1.5        # a third as shadow-style queries
    return rays


@pytest.mark.parametrize("name", SCENES)
def test_every_oracle_hit_on_an_orthotope_lies_inside_its_box(oracle, name):
    from ndt_amd.hip import item_boxes
    fs = golden(name).scene
    ids = orthotopes(fs)
    if not ids:
        pytest.skip("no top-level orthotope in this scene")
    boxes = item_boxes(fs)
    assert boxes is not None
    frame, rows, has = boxes
    assert np.abs(frame @ frame.T - np.eye(fs.dims)).max() < 1e-12          # orthonormal
    assert has[ids].all() and not has[[i for i in range(len(has)) if i not in set(ids)]].any()
    rays = aimed_rays(fs, ids, seed=5)
    obj, hit, _ = oracle.trace(fs, rays)
    checked = 0
    for i in set(obj[obj >= 0].tolist()):
        if i >= len(has) or not has[i]:
            continue
        sel = obj == i
        coord = hit[sel] @ frame.T - rows[i, :, 0]
        worst = (np.abs(coord) - (rows[i, :, 1] - SLACK)).max()
        assert worst <= 0, "item %d: a reference hit lies %g outside the box interior" % (i, worst)
        checked += int(sel.sum())
    assert checked > 15 or len(ids) < 8         # (the zoo has two or three small orthotopes: its rays may all miss)


def test_the_frame_of_a_rotated_cube_is_the_cubes_own():
    """Frame 37 of scenes/hypercube.c turns the cube out of the coordinate axes: the boxes stay as thin as the faces
    (thickness 2 x margin in the directions a face does not span) because the frame turns with the cube."""
    from ndt_amd.hip import item_boxes
    fs = golden("c1_hypercube3d_f37").scene
    frame, rows, has = item_boxes(fs)
    assert np.abs(np.abs(frame) - np.eye(3)).max() > 0.05                    # not the world's axes
    thin = rows[has][:, :, 1].min(axis=1)
    assert (thin < 0.05).all()                                               # every face box has a thin direction


def turned(fs, seed=3, angle=0.12):
    """A copy of the scene with every object turned about the origin by a small rotation in two coordinate planes (bounding
    spheres with them; kd-tree, root box, camera and lights stay -- trace_kd's semantics for that data are what they are, the
    oracle and the device must agree on them)."""
    from ndt_amd import load_scene
    g = fs
    n = g.dims
    q = np.eye(n)
    rng = np.random.default_rng(seed)
    for _ in range(2):
        i, j = rng.choice(n, 2, replace=False)
        r = np.eye(n)
        c, s = np.cos(angle), np.sin(angle)
        r[i, i], r[i, j], r[j, i], r[j, j] = c, -s, s, c
        q = r @ q
    vecs = list(g._vecs)
    done = set()
    for o in g.objects:
        for off, cnt in ((o["pos_off"], o["n_pos"]), (o["dir_off"], o["n_dir"]), (o["bounds_center_off"], 1)):
            for k in range(cnt):
                at = off + k * n
                if at < 0 or at in done:
                    continue
                done.add(at)
                vecs[at:at + n] = (q @ np.array(vecs[at:at + n])).tolist()
    g._vecs = vecs
    g._struct = None
    g.finalize()
    return g


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kat_hypercube6d", "kat_hypercube7d"])
def test_boxes_change_nothing_on_the_device(oracle, name):
    from ndt_amd import load_scene
    from ndt_amd.hip import NdtHip, item_boxes
    g = golden(name)
    gpu = NdtHip(0)
    try:
        for variant in ("as it is", "turned"):
            fs = g.scene if variant == "as it is" else turned(load_scene(os.path.join(GOLDEN, g.meta["scene_file"])))
            if variant == "turned":
                frame, _, _ = item_boxes(fs)
                assert np.abs(np.abs(frame) - np.eye(fs.dims)).max() > 0.05          # the frame turned with the objects
            rays = np.concatenate([aimed_rays(fs, orthotopes(fs), seed=9, per_target=2), g.data["kat_in"][:2048]])
            want = oracle.trace(fs, rays[:1200])
            answers = []
            for boxes, together in ((1, 1), (0, 1), (1, 0), (0, 0)):
                gpu.set_option("item_boxes", boxes)
                gpu.set_option("leaf_scan", together)
                gpu.upload_scene(fs)
                answers.append(gpu.trace_rays(rays))
                img, st = gpu.render(64, 36, g.depth)
                answers[-1] = answers[-1] + (img, np.array([st.rays_ref_equiv]))
            for other in answers[1:]:
                for a, b in zip(answers[0], other):
                    assert np.array_equal(a, b), variant
            for a, b in zip(answers[0][:3], want):
                assert np.array_equal(a[:1200], b), variant
            assert (answers[0][0] >= 0).mean() > 0.2
    finally:
        gpu.close()
