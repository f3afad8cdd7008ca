"""The hcube hull box (ndt_blob.hip:hcube_hull_box, ndt_hip_hcube_hull_box) is the one place
where the device does LESS than the reference: a ray that misses the box skips the nested
trace() over the hcube's faces (hcube.c:241).  That is only legitimate if no point the
reference can return for a face lies outside the box.  Checked here
  * on the CPU: every hit the oracle reports on an hcube, for rays aimed at its faces, lies
    inside the box with the slack the derivation promises (margin 0.02 vs reach 0.01485);
  * on the GPU: the same rays give bit-identical answers with the oracle, with the boxes
    switched off (ndt_hip_set_option "hull_box" 0) and with the hull box alone ("face_box" 0).
The same holds one face at a time for the per-face boxes: only the faces whose box a ray meets are scanned.
"""
import os

import numpy as np
import pytest

from conftest import golden

HCUBE = 6
SCENES = ["c3_random4d", "zoo4d", "zoo3d_mirror", "zoo5d_f2", "zoo6d"]
SLACK = 0.02 - 0.01485


def hcubes(fs):
    return [i for i, o in enumerate(fs.objects) if o["type"] == HCUBE and o["parent"] < 0]


def aimed_rays(fs, box_of, seed, per_target=6):
    """Rays aimed at the places where faces can be hit: around every face's position (where the
    acceptance region of a skewed face sits), over every face's span, and through the box."""
    rng = np.random.default_rng(seed)
    n = fs.dims
    vecs = np.asarray(fs.vecs, dtype=np.float64).ravel()
    flags = np.asarray(fs.flags)
    refs = np.asarray(fs.obj_refs)
    targets = []
    for h in hcubes(fs):
        o = fs.objects[h]
        for k in range(o["n_obj"]):
            f = fs.objects[refs[o["obj_off"] + k]]
            m = int(flags[f["flag_off"]])
            pos = vecs[f["pos_off"]:f["pos_off"] + n]
            dirs = vecs[f["dir_off"]:f["dir_off"] + m * n].reshape(m, n)
            for _ in range(per_target):
                targets.append(pos + rng.normal(0, 0.006, n))
                targets.append(pos + rng.uniform(0, 1, m) @ dirs + rng.normal(0, 0.004, n))
        box = box_of(h)
        if box is not None:
            ax, c, half = box
            for _ in range(8 * per_target):
                targets.append(np.linalg.solve(ax, c + rng.uniform(-1, 1, n) * half))     # frame coordinates -> world
    targets = np.array(targets)
    d = rng.normal(0, 1, targets.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.uniform(2.0, 12.0, (len(targets), 1))
    rays = np.zeros((len(targets), 2 * n + 1))
    rays[:, :n] = targets - d * dist
    rays[:, n:2 * n] = d
    rays[:, 2 * n] = -1.0
    # a third of them as shadow-style queries (any hit closer than the limit)
    rays[::3, 2 * n] = dist[::3, 0] * 1.5
    return rays


@pytest.mark.parametrize("name", SCENES)
def test_every_oracle_hit_on_an_hcube_lies_inside_its_hull_box(oracle, name):
    from ndt_amd.hip import hcube_hull_box
    fs = golden(name).scene
    hs = hcubes(fs)
    if not hs:
        pytest.skip("no hcube in this scene")
    boxes = {h: hcube_hull_box(fs, h) for h in hs}
    assert any(b is not None for b in boxes.values())
    for b in boxes.values():
        if b is not None:
            ax = b[0]
            # unit covectors: an orthonormal frame, or the dual basis of the cube's edge directions (a slab test needs no more)
            assert np.abs(np.linalg.norm(ax, axis=1) - 1).max() < 1e-12 and abs(np.linalg.det(ax)) > 1e-6
    rays = aimed_rays(fs, boxes.get, seed=11)
    obj, hit, _ = oracle.trace(fs, rays)
    n_hits = 0
    for h in hs:
        sel = obj == h
        n_hits += int(sel.sum())
        if boxes[h] is None or not sel.any():
            continue
        ax, c, half = boxes[h]
        coord = hit[sel] @ ax.T - c
        worst = (np.abs(coord) - (half - SLACK)).max()
        assert worst <= 0, "hcube %d: a reference hit lies %g outside the box interior" % (h, worst)
    assert n_hits > 15          # the aimed rays do reach the faces (skewed ones of random.c included)


@pytest.mark.parametrize("name", SCENES)
def test_every_oracle_hit_on_an_hcube_lies_inside_a_face_box(oracle, name):
    """The per-face boxes (ndt_hip_hcube_face_boxes): the device scans only the faces whose box the ray
    meets, so every hit the reference reports must lie inside the box of a face that can be hit."""
    from ndt_amd.hip import hcube_hull_box, hcube_face_boxes
    fs = golden(name).scene
    hs = hcubes(fs)
    if not hs:
        pytest.skip("no hcube in this scene")
    boxes = {h: hcube_hull_box(fs, h) for h in hs}
    faces = {h: hcube_face_boxes(fs, h) for h in hs}
    if all(f is None for f in faces.values()):
        assert all(boxes[h] is None for h in hs)      # (hcubes of any number of faces get them: 63 to a mask word on the device)
        pytest.skip("no hcube of this scene has face boxes")
    rays = aimed_rays(fs, boxes.get, seed=17)
    obj, hit, _ = oracle.trace(fs, rays)
    checked = 0
    for h in hs:
        sel = obj == h
        if faces[h] is None or not sel.any():
            continue
        ax = boxes[h][0]
        centre, half, live = faces[h]
        assert len(centre) == fs.objects[h]["n_obj"]
        # the hull box is the union of the face boxes
        lo, hi = (centre - half)[live].min(axis=0), (centre + half)[live].max(axis=0)
        assert np.allclose(0.5 * (lo + hi), boxes[h][1], atol=1e-12) and np.allclose(0.5 * (hi - lo), boxes[h][2], atol=1e-12)
        coord = hit[sel] @ ax.T                                         # [hits, N] in the frame
        inside = (np.abs(coord[:, None, :] - centre[None]) <= (half - SLACK)[None]).all(axis=2) & live[None]
        assert inside.any(axis=1).all(), "hcube %d: a reference hit lies in no face box" % h
        checked += int(sel.sum())
    assert checked > 15


@pytest.mark.parametrize("name", SCENES)
def test_face_tree_runs_hold_their_faces(oracle, name):
    """The hierarchy over the face boxes of an hcube of more than 63 faces (ndt_hip_hcube_face_tree): the box of every aligned
    run of 2^j faces holds the boxes of its faces that can be hit -- so a ray that misses a run misses them all -- and every
    hit the reference reports lies inside the run of every level that its face belongs to."""
    from ndt_amd.hip import hcube_hull_box, hcube_face_boxes, hcube_face_tree
    fs = golden(name).scene
    big = [h for h in hcubes(fs) if fs.objects[h]["n_obj"] > 63]
    if not big:
        pytest.skip("no hcube of more than 63 faces in this scene")
    boxes = {h: hcube_hull_box(fs, h) for h in big}
    checked = 0
    for h in big:
        faces, tree = hcube_face_boxes(fs, h), hcube_face_tree(fs, h)
        if faces is None:
            assert tree is None
            continue
        centre, half, live = faces
        nf = len(centre)
        assert len(tree) == int(np.ceil(np.log2(nf)))
        lo, hi = centre - half, centre + half
        for j, (c, hh) in enumerate(tree, start=1):
            assert len(c) == (nf + (1 << j) - 1) >> j
            for k in range(len(c)):
                members = np.flatnonzero(live[k << j:(k + 1) << j]) + (k << j)
                if len(members) == 0:
                    assert (hh[k] < 0).all()
                    continue
                assert (c[k] - hh[k] <= lo[members].min(axis=0)).all() and (c[k] + hh[k] >= hi[members].max(axis=0)).all()
        checked += 1
    assert checked > 0
    # the reference's hits, in every run above their face
    rays = aimed_rays(fs, boxes.get, seed=31)
    obj, hit, _ = oracle.trace(fs, rays)
    n_in = 0
    for h in big:
        sel = obj == h
        faces, tree = hcube_face_boxes(fs, h), hcube_face_tree(fs, h)
        if faces is None or not sel.any():
            continue
        ax = boxes[h][0]
        centre, half, live = faces
        coord = hit[sel] @ ax.T
        for x in coord:
            inside = np.flatnonzero((np.abs(x[None] - centre) <= (half - SLACK)).all(axis=1) & live)
            assert len(inside) > 0
            ok = False
            for f in inside:        # (the face that was hit is one of them: its runs must all hold the point)
                ok = ok or all((np.abs(x - c[f >> j]) <= hh[f >> j]).all() for j, (c, hh) in enumerate(tree, start=1))
            assert ok
            n_in += 1
    print("%s: %d reference hits on hcubes of more than 63 faces, all inside their runs" % (name, n_in))


@pytest.mark.parametrize("name", SCENES + ["zoo9d"])
def test_face_groups_list_every_face_under_axes_whose_clusters_hold_it(name):
    """The index of an hcube's faces by the hull axes their boxes are thin on (ndt_hip_hcube_face_groups): every face that can
    be hit is in the (ascending) list of its axis set, and on every axis of that set its interval lies inside one of the axis's
    two cluster intervals -- so a ray that meets the face's box inside the hull passes a cluster on each of those axes, which is
    all the device's lookup relies on (it tests every listed face against its own box)."""
    from ndt_amd.hip import hcube_face_boxes, hcube_face_groups
    fs = golden(name).scene
    big = [h for h in hcubes(fs) if fs.objects[h]["n_obj"] > 63]
    if not big:
        pytest.skip("no hcube of more than 63 faces in this scene")
    for h in big:
        faces, groups = hcube_face_boxes(fs, h), hcube_face_groups(fs, h)
        if faces is None:
            assert groups is None
            continue
        centre, half, live = faces
        clusters, table, face_set, members = groups
        n = fs.dims
        assert len(face_set) == len(centre)
        assert ((face_set < 0) == ~live).all()
        lo, hi = centre - half, centre + half
        c_lo, c_hi = clusters[:, :, 0] - clusters[:, :, 1], clusters[:, :, 0] + clusters[:, :, 1]       # [axis, side]
        c_hi = np.where(clusters[:, :, 1] < 0, -np.inf, c_hi)
        pinned = 0
        for f in np.flatnonzero(live):
            s = int(face_set[f])
            mine = members[table[s, 0]:table[s, 0] + table[s, 1]]
            assert f in mine and (np.diff(mine) > 0).all()
            for a in range(n):
                if s >> a & 1:
                    assert ((c_lo[a] <= lo[f, a]) & (hi[f, a] <= c_hi[a])).any()
                    pinned += 1
        # an m-face of a cube is pinned on N - m >= 1 axes: the index has something to select by
        assert pinned >= int(live.sum())
        print("%s: hcube %d: %d faces under %d axis sets, %.2f pinned axes a face" % (name, h, int(live.sum()), len(set(face_set[live].tolist())),
                                                                                         pinned / max(1, int(live.sum()))))


@pytest.mark.parametrize("name", SCENES + ["zoo9d"])
def test_hull_frame_is_the_dual_basis_of_a_parallelotopes_edges(name):
    """When the faces of an hcube span exactly N distinct directions (a parallelotope: ndt's hcube, sheared or not), the hull's
    axes are the normalised dual basis of those directions -- axis_k . d_j = 0 for every direction but one -- so that a face is
    thin (two margins) on exactly the axes of the directions it does not span, provided its own directions are orthogonal
    (orthotope.intersect accepts only a blob around the position of a skewed face: those boxes are small on every axis)."""
    from ndt_amd.hip import hcube_hull_box, hcube_face_boxes
    fs = golden(name).scene
    n = fs.dims
    vecs = np.asarray(fs.vecs, dtype=np.float64).ravel()
    flags = np.asarray(fs.flags)
    refs = np.asarray(fs.obj_refs)
    checked = 0
    for h in hcubes(fs):
        box = hcube_hull_box(fs, h)
        if box is None:
            continue
        o = fs.objects[h]
        dirs, faces = [], []
        for k in range(o["n_obj"]):
            f = fs.objects[refs[o["obj_off"] + k]]
            m = int(flags[f["flag_off"]])
            d = vecs[f["dir_off"]:f["dir_off"] + m * n].reshape(m, n)
            d = d / np.linalg.norm(d, axis=1, keepdims=True)
            mine = []
            for u in d:
                j = next((j for j, w in enumerate(dirs) if abs(abs(u @ w) - 1) < 1e-9), None)
                if j is None:
                    dirs.append(u)
                    j = len(dirs) - 1
                mine.append(j)
            faces.append((mine, d))
        if len(dirs) != n or abs(np.linalg.det(np.array(dirs))) < 1e-3:
            continue                                    # not a parallelotope: one of the orthonormal candidate frames
        ax = box[0]
        cross = np.abs(ax @ np.array(dirs).T)           # [axis, direction]
        owner = cross.argmax(axis=1)
        assert sorted(owner.tolist()) == list(range(n))                 # every axis belongs to one direction ...
        off = cross.copy()
        off[np.arange(n), owner] = 0
        assert off.max() < 1e-9                                         # ... and is orthogonal to all the others
        fb = hcube_face_boxes(fs, h)
        if fb is not None:
            centre, half, live = fb
            axis_of = {int(owner[a]): a for a in range(n)}
            for k, (mine, d) in enumerate(faces):
                gram = d @ d.T
                if not live[k] or np.abs(gram - np.eye(len(d))).max() > 1e-9:
                    continue                            # (a skewed face: a blob)
                for j in range(n):
                    if j not in mine:
                        assert half[k, axis_of[j]] <= 0.02 + 2.2e-4 + 1e-12, (k, j, half[k, axis_of[j]])
        checked += 1
    if checked == 0:
        pytest.skip("no parallelotope hcube with a hull box in this scene")


def test_hull_box_of_a_non_hcube_is_an_error():
    from ndt_amd.hip import hcube_hull_box, NdtHipError
    fs = golden("c3_random4d").scene
    other = next(i for i, o in enumerate(fs.objects) if o["type"] != HCUBE)
    with pytest.raises(NdtHipError):
        hcube_hull_box(fs, other)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES + ["zoo9d"])         # (9-D: the hcube of 16 866 faces, the one the index by thin axes is for)
def test_cull_changes_nothing_on_the_device(oracle, name):
    from ndt_amd.hip import NdtHip, hcube_hull_box
    fs = golden(name).scene
    if not hcubes(fs):
        pytest.skip("no hcube in this scene")
    rays = aimed_rays(fs, lambda h: hcube_hull_box(fs, h), seed=23, per_target=12)
    want = oracle.trace(fs, rays)
    gpu = NdtHip(0)
    try:
        gpu.upload_scene(fs)
        got = gpu.trace_rays(rays)
        plain = hull_only = got
        if fs.dims < 9:                                 # (a 9-D upload derives 16 866 face boxes: the variants below suffice there)
            gpu.set_option("hull_box", 0)
            gpu.upload_scene(fs)
            plain = gpu.trace_rays(rays)
            gpu.set_option("hull_box", 1)
            gpu.set_option("face_box", 0)               # hull box only, every face scanned
            gpu.upload_scene(fs)
            hull_only = gpu.trace_rays(rays)
            gpu.set_option("face_box", 1)
        gpu.set_option("face_tree", 0)                  # face boxes walked linearly, 63 at a time (hcubes of more than 63 faces)
        no_tree = None
        if fs.dims < 9:
            gpu.upload_scene(fs)
            no_tree = gpu.trace_rays(rays)
        gpu.set_option("face_groups", 0)                # ... neither the hierarchy nor the index by thin axes
        gpu.upload_scene(fs)
        linear = gpu.trace_rays(rays)
        gpu.set_option("face_tree", 1)                  # the hierarchy alone (round 4's first version)
        gpu.upload_scene(fs)
        tree_only = gpu.trace_rays(rays)
        no_tree = linear if fs.dims >= 9 else no_tree
        gpu.set_option("face_groups", 1)
    finally:
        gpu.close()
    for a, b, c, d, e, f, g in zip(got, plain, want, hull_only, no_tree, linear, tree_only):
        assert np.array_equal(a, b)
        assert np.array_equal(a, c)
        assert np.array_equal(a, d)
        assert np.array_equal(a, e)
        assert np.array_equal(a, f)
        assert np.array_equal(a, g)


def axis_rays(fs, seed):
    """Rays the slab arithmetic finds hardest: along the axes of every hull box's frame (u_k.v is 1 for one slab and a
    rounding error -- 1e-17, its reciprocal 1e17 -- or exactly 0 for the others), through the places faces can be hit and
    just outside the box; rays that start inside the box; and rays that graze a slab's surface at a shallow angle."""
    from ndt_amd.hip import hcube_hull_box, hcube_face_boxes
    rng = np.random.default_rng(seed)
    n = fs.dims
    rays = []
    for h in hcubes(fs):
        box = hcube_hull_box(fs, h)
        if box is None:
            continue
        ax, c, half = box
        faces = hcube_face_boxes(fs, h)
        spots = [c + rng.uniform(-1.02, 1.02, n) * half for _ in range(40)]
        if faces is not None:
            centre, fhalf, live = faces
            spots += [centre[f] + rng.uniform(-1, 1, n) * fhalf[f] for f in np.flatnonzero(live)[:24]]
        back = np.linalg.inv(ax)
        for s in spots:
            p = np.linalg.solve(ax, s)                                      # frame coordinates -> world
            for k in range(n):
                for sign in (1.0, -1.0):
                    d = sign * back[:, k] / np.linalg.norm(back[:, k])      # moves coordinate k only (ax[k] itself in an orthonormal frame)
                    rays.append(np.concatenate([p - d * rng.uniform(3, 9), d, [-1.0]]))
                    rays.append(np.concatenate([p, d, [-1.0]]))             # starts where it aims
                    g = d + 1e-3 * rng.normal(0, 1, n)                      # shallow against the other slabs
                    g /= np.linalg.norm(g)
                    rays.append(np.concatenate([p - g * rng.uniform(3, 9), g, [rng.uniform(2, 20)]]))
        # exactly along the world's axes too (dot products with exact zeros)
        for k in range(n):
            e = np.zeros(n)
            e[k] = 1.0
            p = np.linalg.solve(ax, c + rng.uniform(-1, 1, n) * half)
            rays.append(np.concatenate([p - 5 * e, e, [-1.0]]))
    return np.array(rays)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_rays_along_the_frame_axes_and_from_inside_the_box(oracle, name):
    """The hull test takes 1/(u_k.v) from the hardware's reciprocal (ndt_device.hpp:slab_rcp) and lets its comparisons
    give by 2^-30: the extreme quotients must still never cull what the reference hits."""
    from ndt_amd.hip import NdtHip
    fs = golden(name).scene
    if not hcubes(fs):
        pytest.skip("no hcube in this scene")
    rays = axis_rays(fs, seed=29)
    if len(rays) == 0:
        pytest.skip("no hcube of this scene has a hull box")
    want = oracle.trace(fs, rays)
    gpu = NdtHip(0)
    try:
        gpu.upload_scene(fs)
        got = gpu.trace_rays(rays)
        gpu.set_option("hull_box", 0)
        gpu.upload_scene(fs)
        plain = gpu.trace_rays(rays)
    finally:
        gpu.close()
    for a, b, c in zip(got, plain, want):
        assert np.array_equal(a, b)
        assert np.array_equal(a, c)
    assert (want[0] >= 0).sum() > 10
