"""Parity of the HIP path (libndt_hip.so, through its C ABI) with the oracle and the golden
fixtures.  Needs a real MI355X: run with `pytest -m gpu`.

Tolerance: BASELINE.json's north_star asks per-pixel RGB delta < 1e-4 on the linear double
framebuffer.  The device uses the same IEEE +,-,*,/,sqrt sequence as the reference (no FMA),
so geometry is bit-identical; only acos/cos/sin/asin/pow come from ocml instead of glibc and
may differ in the last ulps, hence TOL_TIGHT is what we actually expect and TOL_SPEC the bar.
"""
import os

import numpy as np
import pytest

from conftest import (comparable, golden, GOLDEN, SMALL_CASES, KAT_CASES, FULL_CASES, AA_CASES, VIEW_CASES, SAMPLED_CASES,
                      SAMPLED_MODE_CASES)

pytestmark = pytest.mark.gpu

TOL_SPEC = 1e-4      # north_star tolerance (linear RGB)
TOL_TIGHT = 1e-9     # what differing libm ulps can explain


@pytest.fixture(scope="module")
def gpu():
    from ndt_amd.hip import NdtHip
    ctx = NdtHip(0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", SMALL_CASES)
def test_framebuffer_vs_reference_golden(gpu, name):
    g = golden(name)
    gpu.upload_scene(g.scene)
    out, st = gpu.render(g.width, g.height, g.depth)
    ref = g.data["fb"]
    diff = np.abs(out - ref)
    assert diff.max() < TOL_SPEC, "max abs diff %g" % diff.max()
    assert (diff > TOL_TIGHT).sum() == 0, "%d values differ by more than %g (max %g)" % (
        (diff > TOL_TIGHT).sum(), TOL_TIGHT, diff.max())
    # ray accounting: the reference's own trace_kd call count
    assert st.rays_ref_equiv == g.meta["rays_total"]
    assert st.rays_primary == g.width * g.height


@pytest.mark.parametrize("name", SMALL_CASES)
def test_framebuffer_vs_oracle(gpu, oracle, name):
    g = golden(name)
    gpu.upload_scene(g.scene)
    # a different size than the fixture, not a multiple of the 8x8 tile
    w, h = g.width - 3, g.height - 5
    out, st = gpu.render(w, h, g.depth)
    want, so = oracle.render(g.scene, w, h, g.depth)
    diff = np.abs(out - want)
    assert diff.max() < TOL_TIGHT, "max abs diff %g" % diff.max()
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv) == (
        so.rays_primary, so.rays_secondary, so.rays_shadow, so.rays_ref_equiv)


@pytest.mark.parametrize("name", KAT_CASES)
def test_trace_kd_known_answers(gpu, name):
    """trace_kd on the device vs the reference's answers: geometry has no libm in it for
    every type but facet (acos), so object ids match exactly and points to the last bit."""
    g = golden(name)
    gpu.upload_scene(g.scene)
    rays, want = g.data["kat_in"], g.data["kat_out"]
    d = g.scene.dims
    obj, hit, nrm = gpu.trace_rays(rays)
    assert np.array_equal(obj, want[:, 1].astype(np.int32))
    assert np.array_equal(hit, want[:, 2:2 + d])
    assert np.array_equal(nrm, want[:, 2 + d:2 + 2 * d])


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_resolution_8bit_vs_reference(gpu, name):
    """Every BASELINE.json config at its stated size (configs[0] 256x256, [1]/[2] 1920x1080, [3] 3840x2160, [4] 6-D..8-D
    at 1920x1080) against the bytes the reference writes; the observed mismatch count is printed (-s shows it)."""
    import ctypes as C
    g = golden(name)
    gpu.upload_scene(g.scene)
    out, st = gpu.render(g.width, g.height, g.depth)
    got = (np.sqrt(np.maximum(0.0, np.minimum(1.0, out))) * 255).astype(np.uint8)     # pixel_d2c, image.h:36
    ref = g.data["rgba8"]
    mism = (got != ref)
    print("%s: %d of %d bytes differ from the reference's image" % (name, int(mism.sum()), mism.size))
    # a last-ulp libm difference can only move a byte when sqrt(x)*255 sits on an integer
    assert mism.sum() <= 16, "%d byte mismatches" % mism.sum()
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1
    assert st.rays_ref_equiv == g.meta["rays_total"]


def test_row_shards_assemble_the_full_frame(gpu):
    g = golden("c3_random4d")
    gpu.upload_scene(g.scene)
    full, sf = gpu.render(g.width, g.height, g.depth)
    total = 0
    for step in (2, 8):
        for begin in range(step):
            part, sp = gpu.render(g.width, g.height, g.depth, row_begin=begin, row_step=step)
            assert np.array_equal(part, full[begin::step])
            if step == 8:
                total += sp.rays_ref_equiv
    assert total == sf.rays_ref_equiv


def test_one_frame_over_several_contexts_is_the_single_render(gpu):
    """ndt_hip_render_multi: N contexts (here all on the one GPU; one per GPU on a node), rows dealt cyclically like the
    reference's MPI_MODE_ROW (ndt.c:812-820), every context pushing its rows into the assembled frame on the first
    context's device.  The frame -- in doubles and as the stored bytes -- must be the single-context render exactly, for a
    height N does not divide, for a shard of a frame, and with recursive anti-aliasing."""
    import torch
    from ndt_amd.hip import NdtHip, render_multi, IMAGE_F64, IMAGE_RGBA8, MULTI_LOCAL, MULTI_STAGED
    g = golden("c3_random4d")
    gpu.upload_scene(g.scene)
    w, h = g.width, g.height + 1                 # 73 rows
    full, sf = gpu.render(w, h, g.depth)
    full8, _ = gpu.render_rgba8(w, h, g.depth)
    assert np.array_equal(full8, (np.sqrt(np.maximum(0.0, np.minimum(1.0, full))) * 255).astype(np.uint8))
    others = [NdtHip(0) for _ in range(4)]
    try:
        for c in others:
            c.upload_scene(g.scene)
        for n in (2, 3, 5):
            ctxs = [gpu] + others[:n - 1]
            out, st = render_multi(ctxs, w, h, g.depth, IMAGE_F64)
            assert np.array_equal(out, full), n
            assert (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv) == (
                sf.rays_primary, sf.rays_secondary, sf.rays_shadow, sf.rays_ref_equiv)
            out8, _ = render_multi(ctxs, w, h, g.depth, IMAGE_RGBA8)
            assert np.array_equal(out8, full8), n
            assert [c.multi_path_taken() for c in ctxs] == [MULTI_LOCAL] * n      # one GPU: plain stores
        # the path a GPU without peer access takes -- one hipMemcpyPeerAsync into the context's staging buffer on the first
        # device, pushed from there on a stream of its own -- forced ("multi_path" 2; a same-device peer copy is legal), in
        # both formats, with the staging buffer growing (a small frame first) and being reused
        try:
            for c in others:
                c.set_option("multi_path", 2)
            for n, (ww, hh) in ((3, (40, 22)), (5, (w, h)), (2, (w, h))):
                ctxs = [gpu] + others[:n - 1]
                one, _ = gpu.render(ww, hh, g.depth)
                out, st = render_multi(ctxs, ww, hh, g.depth, IMAGE_F64)
                assert np.array_equal(out, one), n
                assert [c.multi_path_taken() for c in ctxs] == [MULTI_LOCAL] + [MULTI_STAGED] * (n - 1)
                out8, _ = render_multi(ctxs, ww, hh, g.depth, IMAGE_RGBA8)
                assert np.array_equal(out8, (np.sqrt(np.maximum(0.0, np.minimum(1.0, one))) * 255).astype(np.uint8)), n
            gpu.set_option("multi_path", 2)         # ... and the first context staging into itself
            out, _ = render_multi([gpu] + others[:2], w, h, g.depth, IMAGE_F64)
            assert np.array_equal(out, full) and gpu.multi_path_taken() == MULTI_STAGED
        finally:
            gpu.set_option("multi_path", 0)
            for c in others:
                c.set_option("multi_path", 0)
        ctxs = [gpu] + others[:2]
        part, _ = render_multi(ctxs, w, h, g.depth, IMAGE_F64, row_begin=1, row_step=2)     # a shard of the frame, split again
        assert np.array_equal(part, full[1::2])
        aa_one, sa = gpu.render(w, h, g.depth, aa=(12, 2))
        aa_multi, sm = render_multi(ctxs, w, h, g.depth, IMAGE_F64, aa=(12, 2))
        assert np.array_equal(aa_multi, aa_one) and sm.pixels_resampled == sa.pixels_resampled
        # the assembled frame left in HBM on the first context's device
        dev = torch.full((h, w, 4), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        render_multi(ctxs, w, h, g.depth, IMAGE_F64, d_out_ptr=dev.data_ptr())
        assert np.array_equal(dev.cpu().numpy(), full)
    finally:
        for c in others:
            c.close()


def test_one_frame_over_two_gpus(gpu):
    """The cross-device branches of ndt_hip_render_multi, on boxes that have a second GPU (skipped on the one-GPU box, where the
    staged path is driven between contexts of one device above): peer stores over xGMI and the forced staged copy must both
    assemble the single-GPU frame, in doubles and in bytes."""
    from ndt_amd.hip import NdtHip, load_library, render_multi, IMAGE_F64, IMAGE_RGBA8, MULTI_LOCAL, MULTI_PEER, MULTI_STAGED
    if load_library().ndt_hip_device_count() < 2:
        pytest.skip("one GPU")
    g = golden("c3_random4d")
    gpu.upload_scene(g.scene)
    w, h = g.width, g.height + 1
    full, _ = gpu.render(w, h, g.depth)
    full8 = (np.sqrt(np.maximum(0.0, np.minimum(1.0, full))) * 255).astype(np.uint8)
    other = NdtHip(1)
    try:
        other.upload_scene(g.scene)
        for mode, want in ((0, (MULTI_PEER, MULTI_STAGED)), (2, (MULTI_STAGED,))):
            other.set_option("multi_path", mode)
            for _ in range(2):
                out, _ = render_multi([gpu, other], w, h, g.depth, IMAGE_F64)
                assert np.array_equal(out, full), mode
                assert gpu.multi_path_taken() == MULTI_LOCAL and other.multi_path_taken() in want
                out8, _ = render_multi([gpu, other], w, h, g.depth, IMAGE_RGBA8)
                assert np.array_equal(out8, full8), mode
    finally:
        other.close()


@pytest.mark.parametrize("name", ["c3_random4d", "zoo3d_mirror", "c1_hypercube3d", "c5_hypercube6d", "zoo4d"])
def test_every_pipeline_renders_the_same_frame(gpu, name):
    """The three ways a pass can be rendered -- one trace launch + shade launches per bounce (levels), the whole ray tree
    in one persistent launch (stream), the first two bounces per bounce and the deeper ones by the frame kernel (hybrid) --
    differ in which wavefront computes what and when, never in an operand: the images are the same to the last bit, and so
    are the ray counts.  (zoo3d_mirror: facing mirrors, 60 bounces deep; the pools are regrown on the way.)"""
    g = golden(name)
    gpu.upload_scene(g.scene)
    outs = []
    try:
        for pipeline, fused in ((1, 1), (2, 1), (2, 0), (3, 1)):
            # (stream_fused 0: the frame kernel between k_primary and k_finish_pixels instead of doing their work itself)
            gpu.set_option("pipeline", pipeline)
            gpu.set_option("stream_fused", fused)
            img, st = gpu.render(g.width, g.height, g.depth)
            outs.append((img, (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv, st.levels)))
    finally:
        gpu.set_option("pipeline", 0)
        gpu.set_option("stream_fused", 1)
    for img, counts in outs[1:]:
        assert np.array_equal(img, outs[0][0])
        assert counts == outs[0][1]
    assert np.abs(outs[0][0] - g.data["fb"]).max() < TOL_TIGHT


@pytest.mark.parametrize("name", ["c3_random4d", "zoo4d", "c1_hypercube3d"])
def test_item_sets_change_nothing(gpu, oracle, name):
    """Scenes of up to 64 items whose leaf lists are ascending (what the reference's kd builder produces) are traced with
    64-bit item sets per leaf instead of the lists (ndt_device.hpp:trace_kd).  Same answers and pixels as with the lists
    ("item_sets" 0); and a scene whose leaf lists are NOT ascending must not take the set kernels: its answers are the
    oracle's for that order."""
    from ndt_amd import load_scene
    g = golden(name)
    rays = g.data["kat_in"] if "kat_in" in g.data else None
    try:
        gpu.upload_scene(g.scene)
        img1, st1 = gpu.render(g.width, g.height, g.depth)
        kat1 = gpu.trace_rays(rays) if rays is not None else None
        gpu.set_option("item_sets", 0)
        gpu.upload_scene(g.scene)
        img0, st0 = gpu.render(g.width, g.height, g.depth)
        kat0 = gpu.trace_rays(rays) if rays is not None else None
    finally:
        gpu.set_option("item_sets", 1)
    assert np.array_equal(img1, img0)
    assert (st1.rays_primary, st1.rays_secondary, st1.rays_shadow, st1.rays_ref_equiv) == (
        st0.rays_primary, st0.rays_secondary, st0.rays_shadow, st0.rays_ref_equiv)
    if rays is None:
        return
    for a, b in zip(kat1, kat0):
        assert np.array_equal(a, b)
    # 60 000 more rays between random points of the scene's box (closest-hit, directional-shadow and limited queries mixed):
    # sets against lists, and a slice of them against the oracle
    rng = np.random.default_rng(99)
    d = g.scene.dims
    vecs = np.asarray(g.scene.vecs, dtype=np.float64).ravel()
    st_ = g.scene.struct
    lo, hi = vecs[st_.bb_lower_off:st_.bb_lower_off + d], vecs[st_.bb_upper_off:st_.bb_upper_off + d]
    grow = 0.25 * (hi - lo) + 1.0
    a_pts = rng.uniform(lo - grow, hi + grow, (60000, d))
    b_pts = rng.uniform(lo, hi, (60000, d))
    dirs = b_pts - a_pts
    dist = np.linalg.norm(dirs, axis=1, keepdims=True)
    many = np.zeros((60000, 2 * d + 1))
    many[:, :d] = a_pts
    many[:, d:2 * d] = dirs / dist
    many[:, 2 * d] = -1.0
    many[1::3, 2 * d] = 0.0
    many[2::3, 2 * d] = dist[2::3, 0] * rng.uniform(0.3, 1.2, dist[2::3, 0].shape)
    try:
        gpu.upload_scene(g.scene)
        with_sets = gpu.trace_rays(many)
        gpu.set_option("item_sets", 0)
        gpu.upload_scene(g.scene)
        with_lists = gpu.trace_rays(many)
    finally:
        gpu.set_option("item_sets", 1)
    for a, b in zip(with_sets, with_lists):
        assert np.array_equal(a, b)
    want = oracle.trace(g.scene, many[:3000])
    assert np.array_equal(with_sets[0][:3000], want[0])
    assert np.array_equal(with_sets[1][:3000], want[1])
    assert (with_sets[0] >= 0).mean() > 0.05            # (the rays do meet the scene)
    # every leaf list reversed: another scan order, other answers where it matters (ties, the dist_limit break)
    fs = load_scene(os.path.join(GOLDEN, g.meta["scene_file"]))     # a copy of its own: the cached scene stays as it is
    refs = list(fs.leaf_refs)
    for k in fs.kd_nodes:
        if k["dim"] < 0:
            refs[k["first"]:k["first"] + k["num"]] = refs[k["first"]:k["first"] + k["num"]][::-1]
    fs.leaf_refs = refs
    fs._struct = None
    fs.finalize()
    want = oracle.trace(fs, rays)
    gpu.upload_scene(fs)
    got = gpu.trace_rays(rays)
    assert np.array_equal(got[0], want[0])
    assert np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2])


COOP_CASES = ["c3_random4d", "zoo4d", "zoo3d_mirror", "c1_hypercube3d", "zoo5d_f2", "zoo6d"]


@pytest.mark.parametrize("name", COOP_CASES)
def test_cooperative_stragglers_change_nothing(gpu, name):
    """Item-set scenes: the rays a batch of the trace kernel gives up are traced again, one ray per wavefront, by
    coop_trace (every item against the ray side by side, then a scalar replay of trace() / kd_node_intersect).  Forced for
    EVERY ray (budget 0, any number of rays left), for the last eight of every batch, and switched off: the reference's
    trace_kd answers bit for bit, and the same frame -- pixels and ray counts -- as the default settings give."""
    g = golden(name)
    gpu.upload_scene(g.scene)
    rays, want = g.data["kat_in"], g.data["kat_out"]
    d = g.scene.dims
    try:
        gpu.set_option("pipeline", 1)           # the per-bounce kernels: the trace kernel that has the straggler ring
        frames = {}
        for label, (on, budget, live) in {"default": (1, 25, 8), "all": (1, 0, 64), "last8": (1, 0, 8), "off": (0, 25, 8)}.items():
            gpu.set_option("coop", on)
            gpu.set_option("coop_budget_us", budget)
            gpu.set_option("coop_max_live", live)
            obj, hit, nrm = gpu.trace_rays(rays)
            assert np.array_equal(obj, want[:, 1].astype(np.int32)), label
            assert np.array_equal(hit, want[:, 2:2 + d]), label
            assert np.array_equal(nrm, want[:, 2 + d:2 + 2 * d]), label
            out, st = gpu.render(g.width, g.height, g.depth)
            frames[label] = (out, (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv))
        for label in ("all", "last8", "off"):
            assert np.array_equal(frames[label][0], frames["default"][0]), label
            assert frames[label][1] == frames["default"][1], label
        assert np.abs(frames["all"][0] - g.data["fb"]).max() < TOL_TIGHT
    finally:
        gpu.set_option("pipeline", 0)
        gpu.set_option("coop", 0)
        gpu.set_option("coop_budget_us", 25)
        gpu.set_option("coop_max_live", 8)


@pytest.mark.parametrize("name", ["kat_hypercube6d", "kat_hypercube7d", "kat_hypercube8d"])
def test_leaf_history_changes_nothing(gpu, oracle, name):
    """Global-memory tier (6-D .. 8-D hypercubes): a ray's visit mask is kept as up to four {leaf, cut} pairs tested against the
    leaves' item sets (ndt_device.hpp:VisitMask<0>) instead of a bit per item in a 432 MB slab.  The slab alone ("leaf_history"
    0), one or two pairs (every ray that scans a second / third leaf replays its history into the slab and goes on there) and
    four pairs give the same answers -- known answers, 30 000 mixed closest-hit / shadow queries that cross the cube (a slice
    of them against the oracle), and a frame through both pipelines."""
    g = golden(name)
    d = g.scene.dims
    rng = np.random.default_rng(7)
    vecs = np.asarray(g.scene.vecs, dtype=np.float64).ravel()
    st_ = g.scene.struct
    lo, hi = vecs[st_.bb_lower_off:st_.bb_lower_off + d], vecs[st_.bb_upper_off:st_.bb_upper_off + d]
    grow = 0.25 * (hi - lo) + 1.0
    n = 30000
    a_pts = rng.uniform(lo - grow, hi + grow, (n, d))
    b_pts = rng.uniform(lo, hi, (n, d))
    dirs = b_pts - a_pts
    dist = np.linalg.norm(dirs, axis=1, keepdims=True)
    many = np.zeros((n, 2 * d + 1))
    many[:, :d] = a_pts
    many[:, d:2 * d] = dirs / dist
    many[:, 2 * d] = -1.0
    many[1::3, 2 * d] = 0.0
    many[2::3, 2 * d] = dist[2::3, 0] * rng.uniform(0.3, 1.2, dist[2::3, 0].shape)
    results = {}
    try:
        for pairs in (4, 0, 1, 2):
            gpu.set_option("leaf_history", pairs)
            gpu.upload_scene(g.scene)
            kat = gpu.trace_rays(g.data["kat_in"])
            rnd = gpu.trace_rays(many)
            frames = []
            for pipeline in (1, 2):
                gpu.set_option("pipeline", pipeline)
                img, st = gpu.render(96, 54, g.depth)
                frames.append((img, st.rays_ref_equiv))
            gpu.set_option("pipeline", 0)
            results[pairs] = (kat, rnd, frames)
    finally:
        gpu.set_option("pipeline", 0)
        gpu.set_option("leaf_history", 4)
    want = g.data["kat_out"]
    assert np.array_equal(results[4][0][0], want[:, 1].astype(np.int32))
    for pairs in (0, 1, 2):
        for a, b in zip(results[4][0], results[pairs][0]):
            assert np.array_equal(a, b), pairs
        for a, b in zip(results[4][1], results[pairs][1]):
            assert np.array_equal(a, b), pairs
        for (ia, ra), (ib, rb) in zip(results[4][2], results[pairs][2]):
            assert np.array_equal(ia, ib) and ra == rb, pairs
    assert np.array_equal(results[4][2][0][0], results[4][2][1][0])
    ow = oracle.trace(g.scene, many[:1500])
    assert np.array_equal(results[4][1][0][:1500], ow[0])
    assert np.array_equal(results[4][1][1][:1500], ow[1])
    assert (results[4][1][0] >= 0).mean() > 0.05


def test_render_is_deterministic(gpu):
    g = golden("c3_random4d")
    gpu.upload_scene(g.scene)
    a, _ = gpu.render(g.width, g.height, g.depth)
    b, _ = gpu.render(g.width, g.height, g.depth)
    assert np.array_equal(a, b)


def test_depth_zero_and_one(gpu, oracle):
    g = golden("c1_hypercube3d")
    gpu.upload_scene(g.scene)
    for depth in (0, 1, 2):
        out, _ = gpu.render(40, 24, depth)
        want, _ = oracle.render(g.scene, 40, 24, depth)
        assert np.abs(out - want).max() < TOL_TIGHT


def test_specular_disabled_path(gpu, oracle):
    g = golden("c3_random4d")
    gpu.upload_scene(g.scene)
    out, _ = gpu.render(64, 36, g.depth, specular=0)
    want, _ = oracle.render(g.scene, 64, 36, g.depth, specular=0)
    assert np.abs(out - want).max() < TOL_TIGHT


@pytest.mark.parametrize("order", [(1, 0, 3, 2), (3, 2, 1, 0), (2, 1, 0, 3)])
def test_lights_in_any_order(gpu, oracle, order):
    """The shadow queue has one segment per light that is not ambient, in list order, and the trace launch finds a segment's
    light again from the scene (the rays of a point or spot light are stored without their origin: ndt_kernels.hip:
    seg_light_origin): with the ambient light in the middle of the list or at its end, and the point / spot / directional
    lights in every relative order, both pipelines must still give the oracle's frame (apply_lights sums in list order,
    ndt.c:98)."""
    from ndt_amd import load_scene
    g = golden("zoo4d")
    fs = load_scene(os.path.join(GOLDEN, "zoo4d.ndtscene.gz"))        # (a copy of its own: the fixture's scene is shared)
    assert [l["type"] for l in fs.lights] == [0, 1, 3, 2]
    fs.lights = [fs.lights[i] for i in order]
    fs._struct = None
    fs.finalize()
    want, wst = oracle.render(fs, 96, 54, g.depth)
    for pipeline in (1, 2):
        gpu.set_option("pipeline", pipeline)
        gpu.upload_scene(fs)
        out, st = gpu.render(96, 54, g.depth)
        assert np.abs(out - want).max() < TOL_TIGHT, "pipeline %d, lights %s" % (pipeline, order)
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv) == (
            wst.rays_primary, wst.rays_secondary, wst.rays_shadow, wst.rays_ref_equiv)
    gpu.set_option("pipeline", 0)


def test_quantize_on_device(gpu, oracle):
    import torch
    g = golden("c2_balls4d")
    gpu.upload_scene(g.scene)
    out, _ = gpu.render(g.width, g.height, g.depth)
    t = torch.from_numpy(out).cuda()
    q = torch.empty(out.shape, dtype=torch.uint8, device="cuda")
    gpu.quantize_device(t.data_ptr(), q.data_ptr(), out.shape[0] * out.shape[1])
    gpu.synchronize()
    assert np.array_equal(q.cpu().numpy(), oracle.quantize(out))


def test_frames_delivered_behind_the_next_render(gpu):
    """ndt_hip_render_rgba8_async: the copy of a frame's bytes to pinned host memory runs behind the next frame's rendering; after
    ndt_hip_render_rgba8_wait every frame of the sequence is the synchronous call's, byte for byte -- different sizes and scenes
    in flight at once, more frames than buffers."""
    import torch
    g, g2 = golden("c3_random4d"), golden("zoo4d")
    frames = [(g, 96, 54), (g, 128, 72), (g2, 64, 36), (g, 40, 22), (g2, 96, 54)]
    bufs = [torch.zeros((h, w, 4), dtype=torch.uint8).pin_memory() for _, w, h in frames]
    sync = {}
    for k, (gg, w, h) in enumerate(frames):
        gpu.upload_scene(gg.scene)
        sync[k] = gpu.render_rgba8(w, h, gg.depth)[0]
    for k, ((gg, w, h), b) in enumerate(zip(frames, bufs)):
        gpu.upload_scene(gg.scene)
        gpu.render_rgba8_async(b.data_ptr(), w, h, gg.depth)
        if k > 0:       # when the call returns, the frame before has arrived
            assert np.array_equal(bufs[k - 1].numpy(), sync[k - 1]), k
    gpu.render_rgba8_wait()
    for (gg, w, h), b in zip(frames, bufs):
        gpu.upload_scene(gg.scene)
        want, _ = gpu.render_rgba8(w, h, gg.depth)
        assert np.array_equal(b.numpy(), want)


def test_bad_scene_is_rejected_not_rendered(gpu):
    from ndt_amd.hip import NdtHipError
    g = golden("c1_hypercube3d")
    st = g.scene.struct
    old = st.cam_type
    st.cam_type = 1          # CAMERA_VR on a scene without the VR fields (an `ndtscene 1` file has no local axes): refused, not guessed
    try:
        with pytest.raises(NdtHipError):
            gpu.upload_scene(g.scene)
    finally:
        st.cam_type = old
    gpu.upload_scene(g.scene)


@pytest.mark.parametrize("name", AA_CASES)
def test_recursive_antialiasing_vs_reference(gpu, name):
    """Whitted's recursive anti-aliasing (-a diff,depth): the device walks the recursion tree level
    by level; the resampled image, the number of resampled pixels and the reference's trace_kd
    count must come out the same."""
    g = golden(name)
    aa = (g.meta["aa_diff"], g.meta["aa_depth"])
    gpu.upload_scene(g.scene)
    if "depth" in g.data:
        # the depth map render_image makes beside an anti-aliased image: the first pass's (ndt.c:930-935, 753-756)
        out, dm, st = gpu.render(g.width, g.height, g.depth, aa=aa, stereo=g.meta.get("stereo", 0), depth_map=True)
        assert np.abs(dm - g.data["depth"]).max() < TOL_TIGHT and (dm > 0).any()
    else:
        out, st = gpu.render(g.width, g.height, g.depth, aa=aa, stereo=g.meta.get("stereo", 0))
    ref = g.data["fb"]
    diff = np.abs(out - ref)
    # (the VR screen goes through sin / cos: ocml against glibc in the primary rays, as in the view cases)
    assert diff.max() < (1e-7 if g.meta.get("config") == "vr" else TOL_TIGHT), "max abs diff %g" % diff.max()
    assert st.pixels_resampled == g.meta["pixels_resampled"]
    # The reference's trace_kd count.  Refracted directions go through acos / sin / asin / cos (vectNd.c:119-200): ocml's differ
    # from glibc's in the last bit, everything behind a refraction is a last-bit different ray, and about one such ray in
    # 10^5 sits on a sign or EPSILON test that then falls the other way -- a shadow ray more or less whose contribution is
    # zero either way (the images agree to 1e-13).  The small fixtures have no such ray (counts identical); the anti-aliased
    # zoo frames, 0.5 - 3 M rays each, have a handful (profiles/aa_stereo_counts.py: mono as well as stereo).
    # So: exact for the fixtures without glass (any counting regression in the fused / streaming paths shows), 1e-5 for the zoo.
    if "zoo" not in name:
        assert st.rays_ref_equiv == g.meta["rays_total"]
    print("%s: device %d reference-equivalent rays, reference %d" % (name, st.rays_ref_equiv, g.meta["rays_total"]))
    assert abs(st.rays_ref_equiv - g.meta["rays_total"]) <= 1e-5 * g.meta["rays_total"], (st.rays_ref_equiv, g.meta["rays_total"])


def test_recursive_antialiasing_row_shards_and_oracle(gpu, oracle):
    g = golden("aa_zoo4d")
    gpu.upload_scene(g.scene)
    aa = (12, 2)
    w, h = g.width + 5, g.height + 2
    want, so = oracle.render(g.scene, w, h, g.depth, aa=aa)
    full, st = gpu.render(w, h, g.depth, aa=aa)
    assert np.abs(full - want).max() < TOL_TIGHT
    assert (st.pixels_resampled, st.aa_samples, st.rays_ref_equiv) == (so.pixels_resampled, so.aa_samples, so.rays_ref_equiv)
    for begin in range(3):
        part, _ = gpu.render(w, h, g.depth, row_begin=begin, row_step=3, aa=aa)
        assert np.array_equal(part, full[begin::3])


def test_antialiasing_cutoffs(gpu, oracle):
    """aa_depth 0 (the recursion's cut-off at the first call) and a threshold nothing exceeds."""
    g = golden("aa_c1_hypercube3d")
    gpu.upload_scene(g.scene)
    for aa in ((20, 0), (255 * 8, 3)):
        want, so = oracle.render(g.scene, g.width, g.height, g.depth, aa=aa)
        out, st = gpu.render(g.width, g.height, g.depth, aa=aa)
        assert np.abs(out - want).max() < TOL_TIGHT
        assert st.pixels_resampled == so.pixels_resampled


@pytest.mark.parametrize("name", VIEW_CASES)
def test_stereo_vr_pano_and_depth_maps_vs_reference(gpu, name):
    """Stereo modes, VR / panorama cameras (sin/cos/tan from ocml instead of glibc: last-ulp
    differences in the primary rays) and depth maps against the compiled reference."""
    g = golden(name)
    stereo = g.meta.get("stereo", 0)
    gpu.upload_scene(g.scene)
    if "depth" in g.data:
        out, dm, st = gpu.render(g.width, g.height, g.depth, stereo=stereo, depth_map=True)
        assert np.abs(dm - g.data["depth"]).max() < TOL_TIGHT
    else:
        out, st = gpu.render(g.width, g.height, g.depth, stereo=stereo)
    diff = np.abs(comparable(out, stereo) - comparable(g.data["fb"], stereo))
    assert diff.max() < (TOL_TIGHT if g.scene.cam_type == 0 else 1e-7), "max abs diff %g" % diff.max()
    assert st.rays_ref_equiv == g.meta["rays_total"]


def test_stereo_row_shards(gpu):
    g = golden("st_zoo4d_ou")
    gpu.upload_scene(g.scene)
    full, _ = gpu.render(g.width, g.height, g.depth, stereo=2)
    for begin in range(2):
        part, dm, _ = gpu.render(g.width, g.height, g.depth, row_begin=begin, row_step=2, stereo=2, depth_map=True)
        assert np.array_equal(part, full[begin::2])
        assert dm.shape == part.shape[:2]


def test_stereo_needs_the_eyes(gpu):
    from ndt_amd.hip import NdtHipError
    g = golden("c1_hypercube3d")            # an `ndtscene 1` file: no eyes
    gpu.upload_scene(g.scene)
    with pytest.raises(NdtHipError):
        gpu.render(32, 32, 4, stereo=1)


_SAMPLER_Z = {}      # (case, S) -> the case's per-value t mean in units of its standard error: ~ N(0, 1) for an unbiased sampler


def _case_seed(name, S, k):
    """A stream-set number of its own for every (case, S, k): cases that share a scene and a size no longer share their draws."""
    import zlib
    return 1 + (zlib.crc32(("%s/%d/%d" % (name, S, k)).encode()) & 0x3fffffff)


def _oracle_ensemble(oracle, g, S, n_seeds):
    """n_seeds renders of the oracle (= the reference's drand48 stream, restarted from n_seeds different states)."""
    s0 = g.meta["seed48"]
    imgs = []
    for k in range(n_seeds):
        img, _ = oracle.render(g.scene, g.width, g.height, g.depth, samples=S, seed48=[(s0[0] + 7919 * k) & 0xffff, s0[1], s0[2]],
                               stereo=g.meta.get("stereo", 0))
        imgs.append(img)
    return np.array(imgs)


@pytest.mark.parametrize("name", SAMPLED_CASES)
def test_jittered_samples_statistically_match_the_oracle(gpu, oracle, name):
    """`-n samples` > 1: jitter + lens sampling + adaptive loop (ndt.c:505-567).  The reference draws from one global
    drand48 stream in pixel order (the oracle follows it exactly, test_oracle_golden.py); the device uses its own
    per-(pixel, sample) streams, so the device's image is ONE MORE DRAW from the distribution the oracle's images are
    drawn from.  Tested against an ensemble of 24 oracle renders from different drand48 states: per value,
    z = (device - ensemble mean) / (ensemble sd * sqrt(1 + 1/24)); the thresholds are what the oracle achieves against
    its own ensemble (edge pixels are bimodal, so |z| has heavier tails than a normal: 99.6 % below 5 was the worst seen),
    and a bias of 0.002 in the image mean -- a fifth of what the first version of this test let through -- fails."""
    g = golden(name)
    gpu.upload_scene(g.scene)
    stereo = g.meta.get("stereo", 0)
    errs = []
    n_seeds = 24
    for S in (8, 128):
        ens = _oracle_ensemble(oracle, g, S, n_seeds)
        out, st = gpu.render(g.width, g.height, g.depth, samples=S, stereo=stereo)
        assert st.aa_samples >= S * g.width * g.height          # at least S samples everywhere
        assert st.rays_primary >= st.aa_samples                 # (samples rendered ahead of the loop's exit are dropped)
        mu, sd = ens.mean(axis=0), ens.std(axis=0, ddof=1)
        noisy = sd > 1e-12
        # values the sampling cannot move (flat background, alpha of covered pixels): equal outright -- but for the odd pixel
        # of which an edge cuts off a sliver that none of the oracle's ~250 samples found and one of the device's did
        # (ns_vr_zoo4d has one: 1 of 18 samples, alpha off by 1/18): at most 3 values in a thousand
        moved = np.abs(out - mu)[~noisy] >= 1e-9
        assert moved.sum() <= 0.003 * moved.size + 1, (int(moved.sum()), moved.size)
        z = (out - mu)[noisy] / (sd[noisy] * np.sqrt(1.0 + 1.0 / n_seeds))
        frac5 = float((np.abs(z) < 5).mean())
        print("%s S=%d: %d noisy values, mean z %+.3f (|z| clipped at 15; %d beyond), %.2f %% |z|<5, max |z| %.1f" % (
            name, S, int(noisy.sum()), np.clip(z, -15, 15).mean(), int((np.abs(z) >= 15).sum()), 100 * frac5, np.abs(z).max()))
        # (a pixel of which an edge cuts off a sliver: its ensemble is constant to 1e-5 until a sample lands in the sliver -- one
        # of the device's eleven did in ns_vr_zoo4d's pixel (5, 12), z = -1263 on all three channels.  Such values are
        # counted, at most 5 in a thousand, and clipped in the bias statistic instead of deciding it alone)
        far = np.abs(z) >= 15
        assert far.sum() <= 0.005 * z.size, (int(far.sum()), z.size)
        assert abs(np.clip(z, -15, 15).mean()) < 0.15, np.clip(z, -15, 15).mean()
        assert frac5 > 0.99
        # unbiased: the image mean against the spread of the ensemble's image means
        means = ens[..., :3].mean(axis=(1, 2, 3))
        t = (out[..., :3].mean() - means.mean()) / (means.std(ddof=1) * np.sqrt(1.0 + 1.0 / n_seeds))
        assert abs(t) < 4.5, t
        assert abs(out[..., :3].mean() - means.mean()) < 0.002
        errs.append(np.abs(out[..., :3] - mu[..., :3]).mean())
        # Two samples of one distribution.  Round 2's twelve `mean z` above were all negative: they are ONE draw, not twelve --
        # every case and every S uses the same default set of streams (same pixel, same sample number: same random numbers),
        # and five of the six cases are the same scene at the same size.  With other stream sets ("sample_seed") the device
        # draws an ensemble of its own: per value t = (device mean - oracle mean) / standard error.  Calibration
        # (profiles/r03_sampler_calibration.txt): oracle against oracle gives mean t within 0 +- 1/sqrt(n) and an rms of
        # 0.99 - 1.07; the device against the oracle (48 draws each) gave -0.040 .. +0.150, six of twelve negative.
        # (round 4: every case and every S draws ITS OWN stream sets -- sample_seed = a hash of (case, S, k) -- so that the
        # twelve statistics of this test are twelve independent draws and their signs mean something: the combined statistic
        # is asserted in test_sampler_statistic_over_all_cases)
        n_dev = 12
        dev = []
        try:
            for k in range(n_dev):
                gpu.set_option("sample_seed", _case_seed(name, S, k))
                dev.append(gpu.render(g.width, g.height, g.depth, samples=S, stereo=stereo)[0])
        finally:
            gpu.set_option("sample_seed", 0)
        dev = np.array(dev)
        se = np.sqrt(ens.var(axis=0, ddof=1) / n_seeds + dev.var(axis=0, ddof=1) / n_dev)
        both = se > 1e-12
        t2 = (dev.mean(axis=0) - mu)[both] / se[both]
        t2 = np.clip(t2, -15, 15)           # (slivers again: a value that is constant in one ensemble and not in the other)
        print("%s S=%d: device ensemble (%d) against oracle ensemble (%d): per-value t mean %+.3f (unbiased: 0 +- %.3f), rms %.2f" % (
            name, S, n_dev, n_seeds, t2.mean(), 1 / np.sqrt(t2.size), np.sqrt((t2 ** 2).mean())))
        # (the three channels of a pixel move together: a third as many independent values)
        _SAMPLER_Z[(name, S)] = float(t2.mean() * np.sqrt(t2.size / 3.0))
        assert abs(t2.mean()) < 4.5 / np.sqrt(t2.size / 3.0), t2.mean()
        assert np.sqrt((t2 ** 2).mean()) < 1.35
        dm = dev[..., :3].mean(axis=(1, 2, 3))
        t_img = (dm.mean() - means.mean()) / np.sqrt(means.var(ddof=1) / n_seeds + dm.var(ddof=1) / n_dev)
        assert abs(t_img) < 4.5, t_img
    # noise, not bias: 16x the samples shrink the error by about sqrt(16) (the adaptive loop takes more than 8 samples
    # where the colour still moves, so somewhat less: the oracle against its own ensemble gives 0.34)
    assert 0.15 < errs[1] / errs[0] < 0.5, errs
    # same call, same image; and sharding does not change it
    again, _ = gpu.render(g.width, g.height, g.depth, samples=8, stereo=stereo)
    first, _ = gpu.render(g.width, g.height, g.depth, samples=8, stereo=stereo)
    assert np.array_equal(again, first)
    part, _ = gpu.render(g.width, g.height, g.depth, samples=8, row_begin=1, row_step=2, stereo=stereo)
    assert np.array_equal(part, first[1::2])


def test_sampler_statistic_over_all_cases():
    """The twelve per-case statistics of the test above (six cases x S = 8, 128), now independent draws: their signs, and their
    sum in units of its standard error.  A sampler with a bias of the size round 3's table hinted at (-0.02 ... -0.06 sigma per
    value, the same sign everywhere) would put |Z| near 10; unbiased, Z ~ N(0, 1)."""
    if len(_SAMPLER_Z) < 4:
        pytest.skip("runs after test_jittered_samples_statistically_match_the_oracle (same process)")
    z = np.array(list(_SAMPLER_Z.values()))
    Z = z.sum() / np.sqrt(z.size)
    print("sampler: %d independent case statistics, %d negative / %d positive, each in sigma: %s; combined Z = %+.2f" % (
        z.size, int((z < 0).sum()), int((z > 0).sum()), " ".join("%+.2f" % v for v in z), Z))
    assert abs(Z) < 3.5, Z
    assert np.abs(z).max() < 4.5, z


@pytest.mark.parametrize("name", SAMPLED_MODE_CASES)
def test_jittered_samples_in_the_other_modes(gpu, oracle, name):
    """-n > 1 as an anaglyph (two adaptive loops per pixel: left eye, then right), frame-packed (1080 + 45 + 1080 lines; the jitter
    spans 1/height of the image, under half a line of an eye's), and with a depth map (every sample overwrites the pixel's depth,
    the last one's stays: ndt.c:362-373).  The oracle reproduces the reference's fixtures bit for bit (test_oracle_golden.py);
    the device is compared with an ensemble of oracle renders like the other sampled cases -- colours per value, and the depth
    map against the spread of the ensemble's depth maps."""
    g = golden(name)
    gpu.upload_scene(g.scene)
    stereo = g.meta.get("stereo", 0)
    S, n_seeds, n_dev = 16, 16, 8
    want_depth = "depth" in g.data
    s0 = g.meta["seed48"]
    ens, ens_d = [], []
    for k in range(n_seeds):
        r = oracle.render(g.scene, g.width, g.height, g.depth, samples=S, seed48=[(s0[0] + 7919 * k) & 0xffff, s0[1], s0[2]],
                          stereo=stereo, depth_map=want_depth)
        ens.append(r[0])
        if want_depth:
            ens_d.append(r[1])
    ens = np.array(ens)
    dev, dev_d = [], []
    try:
        for k in range(n_dev):
            gpu.set_option("sample_seed", k)
            r = gpu.render(g.width, g.height, g.depth, samples=S, stereo=stereo, depth_map=want_depth)
            dev.append(r[0])
            if want_depth:
                dev_d.append(r[1])
            assert r[-1].aa_samples >= S * (g.width * (g.height - (46 if stereo == 4 else 0))) * (2 if stereo == 3 else 1)
    finally:
        gpu.set_option("sample_seed", 0)
    dev = np.array(dev)
    # frame packing: the 45 blank lines + the line between them stay black with alpha 1 (the reference leaves their alpha unset)
    if stereo == 4:
        assert np.array_equal(dev[:, 1080:1126, :, :3], np.zeros_like(dev[:, 1080:1126, :, :3]))
        ens, dev = np.array(ens, copy=True), np.array(dev, copy=True)
        ens[:, 1080:1126, :, 3] = 0.0
        dev[:, 1080:1126, :, 3] = 0.0
    mu = ens.mean(axis=0)
    se = np.sqrt(ens.var(axis=0, ddof=1) / n_seeds + dev.var(axis=0, ddof=1) / n_dev)
    both = se > 1e-12
    # values the sampling cannot move agree outright (but for the odd sliver pixel, as in the test above)
    moved = np.abs(dev.mean(axis=0) - mu)[~both] >= 1e-9
    assert moved.sum() <= 0.003 * moved.size + 1
    t = np.clip((dev.mean(axis=0) - mu)[both] / se[both], -15, 15)
    print("%s: %d noisy values, per-value t mean %+.3f (0 +- %.3f), rms %.2f" % (name, t.size, t.mean(), 1 / np.sqrt(t.size / 3.0),
                                                                                 np.sqrt((t ** 2).mean())))
    assert abs(t.mean()) < 4.5 / np.sqrt(t.size / 3.0) and np.sqrt((t ** 2).mean()) < 1.4
    assert abs(dev[..., :3].mean() - ens[..., :3].mean()) < 0.003
    if want_depth:
        # a pixel's depth is ONE sample's (the last): where the oracle's draws all agree (the whole footprint sees one
        # surface at one distance, or nothing) the device agrees; elsewhere it lies inside the range the footprint offers
        ens_d, dev_d = np.array(ens_d), np.array(dev_d)
        lo, hi = ens_d.min(axis=0), ens_d.max(axis=0)
        flat = (hi - lo) < 1e-12
        if flat.any():
            # (but for the pixel an object covers a few per cent of: sixteen oracle draws whose last samples all missed it say
            # "flat", and one of the device's last samples hits it)
            off = np.abs(dev_d - ens_d[0])[:, flat] >= TOL_TIGHT
            assert off.mean() < 0.02, off.mean()
        slack = 0.25 * (hi - lo) + 1e-9
        inside = (dev_d >= lo - slack) & (dev_d <= hi + slack)
        assert inside.mean() > 0.97, inside.mean()
        assert abs(dev_d.mean() - ens_d.mean()) < 0.05 * abs(ens_d.mean()) + 1e-6
    # shards and repeats
    a, b = gpu.render(g.width, g.height, g.depth, samples=4, stereo=stereo)[0], gpu.render(g.width, g.height, g.depth, samples=4, stereo=stereo)[0]
    assert np.array_equal(a, b)
    part = gpu.render(g.width, g.height, g.depth, samples=4, stereo=stereo, row_begin=1, row_step=2)[0]
    assert np.array_equal(part, a[1::2])


def test_area_lights_make_even_one_sample_stochastic(gpu, oracle):
    """LIGHT_DISK / LIGHT_RECT (ndt.c:116-147): a random point of the light per shading evaluation, so
    with -n 1 the adaptive loop's repeats differ and it keeps sampling until the running mean settles.
    Device and oracle (= the reference's stream) agree statistically and take the same number of samples."""
    g = golden("al_zoo4d")
    gpu.upload_scene(g.scene)
    want, so = oracle.render(g.scene, g.width, g.height, g.depth, samples=1, seed48=g.meta["seed48"])
    out, st = gpu.render(g.width, g.height, g.depth, samples=1)
    n = g.width * g.height
    assert abs(st.aa_samples / n - so.rays_primary / n) < 0.15 * so.rays_primary / n
    assert st.aa_samples > 2 * n                        # more than one pass of the loop everywhere
    assert np.abs(out[..., :3] - want[..., :3]).mean() < 0.02
    assert abs(out[..., :3].mean() - want[..., :3].mean()) < 0.01
    again, _ = gpu.render(g.width, g.height, g.depth, samples=1)
    assert np.array_equal(out, again)
    # under recursive anti-aliasing every sample is such a loop (the stochastic anti-aliased render, below)
    aa_dev, sd = gpu.render(g.width, g.height, g.depth, aa=(20, 2))
    aa_ora, so2 = oracle.render(g.scene, g.width, g.height, g.depth, aa=(20, 2), seed48=g.meta["seed48"])
    assert np.abs(aa_dev[..., :3] - aa_ora[..., :3]).mean() < 0.03
    assert abs(aa_dev[..., :3].mean() - aa_ora[..., :3].mean()) < 0.01
    assert 0.5 * so2.pixels_resampled <= sd.pixels_resampled <= 1.5 * so2.pixels_resampled + 5
    # ... and there is no depth map beside one
    from ndt_amd.hip import NdtHipError
    with pytest.raises(NdtHipError):
        gpu.render(g.width, g.height, g.depth, aa=(20, 2), depth_map=True)


def test_recursive_antialiasing_with_a_lens(gpu, oracle):
    """-a with aperture_radius != 0: the reference samples the lens in this mode too (ndt.c:528), so every anti-aliasing sample
    is get_pixel_color's adaptive loop over lens samples and the subdivision runs on noisy colours.  The oracle reproduces the
    reference's fixture bit for bit from the recorded drand48 state (test_oracle_golden.py); the device draws from its own
    streams: an ensemble of device renders against an ensemble of oracle renders, per value and in the number of pixels
    that were subdivided."""
    g = golden("aa_zoo4d_dof")
    aa = (g.meta["aa_diff"], g.meta["aa_depth"])
    gpu.upload_scene(g.scene)
    s0 = g.meta["seed48"]
    n_ora, n_dev = 12, 8
    ens, res_o = [], []
    for k in range(n_ora):
        img, so = oracle.render(g.scene, g.width, g.height, g.depth, aa=aa, seed48=[(s0[0] + 7919 * k) & 0xffff, s0[1], s0[2]])
        ens.append(img)
        res_o.append(so.pixels_resampled)
    dev, res_d = [], []
    try:
        for k in range(n_dev):
            gpu.set_option("sample_seed", k)
            img, st = gpu.render(g.width, g.height, g.depth, aa=aa)
            dev.append(img)
            res_d.append(st.pixels_resampled)
    finally:
        gpu.set_option("sample_seed", 0)
    ens, dev = np.array(ens), np.array(dev)
    se = np.sqrt(ens.var(axis=0, ddof=1) / n_ora + dev.var(axis=0, ddof=1) / n_dev)
    both = se > 1e-12
    t = np.clip((dev.mean(axis=0) - ens.mean(axis=0))[both] / se[both], -15, 15)
    print("aa_zoo4d_dof: %d noisy values, per-value t mean %+.3f (0 +- %.3f), rms %.2f; pixels resampled: device %s, oracle %s" % (
        t.size, t.mean(), 1 / np.sqrt(t.size / 3.0), np.sqrt((t ** 2).mean()), sorted(res_d), sorted(res_o)))
    assert abs(t.mean()) < 4.5 / np.sqrt(t.size / 3.0) and np.sqrt((t ** 2).mean()) < 1.4
    assert abs(dev[..., :3].mean() - ens[..., :3].mean()) < 0.003
    assert abs(np.mean(res_d) - np.mean(res_o)) < 4 * (np.std(res_o, ddof=1) / np.sqrt(n_ora) + np.std(res_d, ddof=1) / np.sqrt(n_dev)) + 2
    again, _ = gpu.render(g.width, g.height, g.depth, aa=aa)
    first, _ = gpu.render(g.width, g.height, g.depth, aa=aa)
    assert np.array_equal(again, first)
    part, _ = gpu.render(g.width, g.height, g.depth, aa=aa, row_begin=1, row_step=2)
    assert np.array_equal(part, first[1::2])


@pytest.mark.gpu
def test_frames_in_sequence_leave_nothing_behind(gpu, oracle):
    """One context, frame after frame: different sizes, depths, scenes, profiled and not, host buffer and device buffer.
    Every frame resets its own counters on the device and reports through the closing record in host-mapped memory
    (k_frame_init / k_frame_done), so the n-th frame must come out exactly like a first one -- image and statistics."""
    import torch
    a, b = golden("c3_random4d"), golden("c1_hypercube3d")
    first = {}
    plan = [(a, 96, 54, 4, 0), (a, 40, 30, 2, 1), (b, 64, 64, 8, 0), (a, 96, 54, 4, 1), (b, 64, 64, 8, 1),
            (a, 40, 30, 2, 0), (a, 96, 54, 1, 0), (a, 96, 54, 4, 0)]
    for g, w, h, depth, profile in plan:
        gpu.upload_scene(g.scene)
        img, st = gpu.render(w, h, depth, profile=profile)
        dev = torch.full((h, w, 4), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        st_dev = gpu.render_device(dev.data_ptr(), w, h, depth, profile=profile)
        # the call returns when the frame's closing record has arrived: the image must be complete for any other stream
        assert np.array_equal(dev.cpu().numpy(), img)
        key = (g.name, w, h, depth)
        stats = (st.rays_primary, st.rays_secondary, st.rays_shadow, st.rays_ref_equiv, st.levels, st.trace_launches)
        assert stats == (st_dev.rays_primary, st_dev.rays_secondary, st_dev.rays_shadow, st_dev.rays_ref_equiv, st_dev.levels,
                         st_dev.trace_launches)
        if key in first:
            assert np.array_equal(img, first[key][0]) and stats == first[key][1], key
        else:
            first[key] = (img, stats)
            want, wst = oracle.render(g.scene, w, h, depth)
            assert np.abs(img - want).max() < 1e-9
            assert st.rays_ref_equiv == wst.rays_ref_equiv


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads, the roofline object and (N=1) the CPU baseline."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", "hypercube3d"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["rays_traced_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["bound_that_holds"] == "fp64_valu_issue_and_latency"
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mray/s" and c["sample"]
    c1 = d["cpu_baseline_t1"]
    assert c1["cores"] == 1 and c1["value"] > 0 and c1["unit"] == "Mray/s"
    assert d["ms_per_step_host_rgba8"] > 0 and d["ms_per_step_host_rgba8_pipelined"] > 0       # (timings of three steps: no relation asserted)
    assert d["host_rgba8_pipelined_bytes_identical"] is True
    assert d["ms_per_step_two_frames_in_flight"] > 0 and d["two_frames_in_flight_identical"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["rgba8", "f64"])
def test_bench_gather_loop_over_rccl_in_a_world_of_one(fmt):
    """bench.py's N>1 frame loop -- double-buffered asynchronous RCCL gather, the events that free a buffer, rank 0's
    de-interleave beside the next frame -- run through RCCL in a world of one, frame and gather buffers poisoned before
    every frame, the last gathered frame compared byte for byte with a plain render (--verify)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--workload",
           "hypercube3d", "--no-cpu-baseline", "--selftest-gather", "--verify", "--gather", fmt]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "verify: gathered frame == the single-GPU render" in out.stderr
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1


@pytest.mark.gpu
def test_two_ranks_over_gloo_gather_hip_pixels():
    """bench.py's N>1 path with TWO ranks (torch.distributed.run, gloo collectives, both ranks on the one GPU): every
    rank renders its rows of BASELINE configs[3]'s 3840x2160 frame with the HIP kernels, the gather assembles them on
    rank 0, and --verify demands the single-GPU render byte for byte.  The line says "strong"."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--verify", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "verify: gathered frame == the single-GPU render" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["width"] == 3840 and d["config"]["height"] == 2160


@pytest.mark.gpu
def test_node_pool_overflow_renders_again():
    """A frame whose ray tree does not fit the node pool flags it on the device (the frame's closing record carries the
    flag), the host doubles the pool and renders the frame again: same image, same counts, larger pool."""
    from ndt_amd.hip import NdtHip
    g = golden("c1_hypercube3d")            # mirrors: about two secondary rays for every three primaries
    ref = NdtHip(0)
    try:
        ref.upload_scene(g.scene)
        want, wst = ref.render(g.width, g.height, g.depth)
    finally:
        ref.close()
    os.environ["NDT_HIP_TEST_SMALL_POOL"] = "1"
    small = NdtHip(0)
    try:
        small.upload_scene(g.scene)
        got, st = small.render(g.width, g.height, g.depth)
        again, st2 = small.render(g.width, g.height, g.depth)
    finally:
        del os.environ["NDT_HIP_TEST_SMALL_POOL"]
        small.close()
    assert wst.rays_secondary > 256                         # the small pool (primaries + 64 nodes) cannot hold them
    assert np.array_equal(got, want) and np.array_equal(again, want)
    for a in (st, st2):
        assert (a.rays_primary, a.rays_secondary, a.rays_shadow, a.rays_ref_equiv) == \
               (wst.rays_primary, wst.rays_secondary, wst.rays_shadow, wst.rays_ref_equiv)
    assert st.node_capacity >= wst.rays_primary + wst.rays_secondary
