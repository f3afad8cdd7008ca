"""Pin the CPU oracle (oracle/ndt_oracle.c) against outputs of the compiled reference.

The fixtures under tests/golden/ were produced by the reference's own render_image /
trace_kd (tests/golden/make_golden.py).  The oracle has to reproduce them BIT FOR BIT:
same glibc libm, no FMA, SSE lane-pair dot order.
"""
import numpy as np
import pytest

from conftest import (comparable, golden, SMALL_CASES, KAT_CASES, FULL_CASES, AA_CASES, VIEW_CASES, SAMPLED_CASES,
                      SAMPLED_MODE_CASES, AA_LENS_CASES)


@pytest.mark.parametrize("name", SMALL_CASES)
def test_framebuffer_bit_exact(oracle, name):
    g = golden(name)
    out, st = oracle.render(g.scene, g.width, g.height, g.depth)
    ref = g.data["fb"]
    assert out.shape == ref.shape
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    # the reference's trace_kd counter (ld --wrap in oracle/ref_shim.c) vs the oracle's replay of
    # the adaptive re-sampling loop (ndt.c:488)
    assert st.rays_ref_equiv == g.meta["rays_total"]
    assert st.rays_primary == g.width * g.height


@pytest.mark.parametrize("name", ["c1_hypercube3d_f37", "c5_hypercube4d"])
def test_literal_resampling_is_identical(oracle, name):
    """Tracing all k identical samples like the reference does changes nothing but run time."""
    g = golden(name)
    a, sa = oracle.render(g.scene, g.width, g.height, g.depth, literal=False)
    b, sb = oracle.render(g.scene, g.width, g.height, g.depth, literal=True)
    assert np.array_equal(a, b)
    assert sa.rays_ref_equiv == sb.rays_ref_equiv == g.meta["rays_total"]


@pytest.mark.parametrize("name", KAT_CASES)
def test_trace_kd_known_answers(oracle, name):
    g = golden(name)
    rays, want = g.data["kat_in"], g.data["kat_out"]
    d = g.scene.dims
    obj, hit, nrm = oracle.trace(g.scene, rays)
    assert np.array_equal(obj, want[:, 1].astype(np.int32))
    assert np.array_equal((obj >= 0).astype(np.float64), want[:, 0])
    assert np.array_equal(hit, want[:, 2:2 + d])
    assert np.array_equal(nrm, want[:, 2 + d:2 + 2 * d])
    # the vectors must actually exercise hits and misses, and every limit kind
    assert (obj >= 0).any() and (obj < 0).any()
    for lim_kind in (rays[:, 2 * d] < 0, rays[:, 2 * d] == 0, rays[:, 2 * d] > 0):
        assert lim_kind.sum() >= len(rays) // 5


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_resolution_8bit(oracle, name):
    """BASELINE.json configs[1]/[2] at 1920x1080: every byte the reference would write."""
    g = golden(name)
    out, st = oracle.render(g.scene, g.width, g.height, g.depth)
    got = oracle.quantize(out)
    assert np.array_equal(got, g.data["rgba8"])
    assert st.rays_ref_equiv == g.meta["rays_total"]


def test_row_sharding_matches_full_frame(oracle):
    g = golden("c3_random4d")
    full, _ = oracle.render(g.scene, g.width, g.height, g.depth)
    for step in (2, 3):
        for begin in range(step):
            part, _ = oracle.render(g.scene, g.width, g.height, g.depth, row_begin=begin, row_step=step)
            assert np.array_equal(part, full[begin::step])


@pytest.mark.parametrize("name", AA_CASES)
def test_recursive_antialiasing_bit_exact(oracle, name):
    """Whitted's recursive anti-aliasing (-a diff,depth; ndt.c:655-733): the resampled image of the
    reference's own render_line + resample_pixel, its "pixels resampled" count and its trace_kd count."""
    g = golden(name)
    aa = (g.meta["aa_diff"], g.meta["aa_depth"])
    if "depth" in g.data:
        # the depth map render_image makes beside an anti-aliased image: the first pass's (ndt.c:930-935, 753-756)
        out, dm, st = oracle.render(g.scene, g.width, g.height, g.depth, aa=aa, stereo=g.meta.get("stereo", 0), depth_map=True)
        assert np.array_equal(dm, g.data["depth"]) and (dm > 0).any()
    else:
        out, st = oracle.render(g.scene, g.width, g.height, g.depth, aa=aa, stereo=g.meta.get("stereo", 0))
    ref = g.data["fb"]
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    assert st.pixels_resampled == g.meta["pixels_resampled"] > 0
    assert st.rays_ref_equiv == g.meta["rays_total"]


@pytest.mark.parametrize("name", AA_LENS_CASES)
def test_recursive_antialiasing_with_a_lens_bit_exact(oracle, name):
    """-a with aperture_radius != 0: the reference samples the lens in this mode too (ndt.c:528: `recursive_aa != 0 ||
    samples > 1`), one drand48 pair (or more: rejection) per get_pixel_color pass, and the adaptive loop runs until the
    colour settles.  One thread, the stream started where the reference's stood: same numbers, same image, same counts."""
    g = golden(name)
    aa = (g.meta["aa_diff"], g.meta["aa_depth"])
    out, st = oracle.render(g.scene, g.width, g.height, g.depth, aa=aa, seed48=g.meta["seed48"])
    assert np.array_equal(out, g.data["fb"]), "max abs diff %g" % np.abs(out - g.data["fb"]).max()
    assert st.pixels_resampled == g.meta["pixels_resampled"] > 0
    assert st.rays_ref_equiv == g.meta["rays_total"]


def test_recursive_antialiasing_row_shards(oracle):
    g = golden("aa_zoo4d")
    aa = (g.meta["aa_diff"], g.meta["aa_depth"])
    full = g.data["fb"]
    for begin in range(3):
        part, _ = oracle.render(g.scene, g.width, g.height, g.depth, row_begin=begin, row_step=3, aa=aa)
        assert np.array_equal(part, full[begin::3])


@pytest.mark.parametrize("name", VIEW_CASES)
def test_stereo_vr_pano_and_depth_maps(oracle, name):
    """Stereo modes (ndt.c:590-650), VR / panorama screens with the eyes going round the centre
    (camera.c:506-555, ndt.c:519-525) and depth maps (ndt.c:362-373): framebuffer, depth map and
    trace_kd count of the compiled reference.  The spherical / cylindrical screens go through
    sin/cos/tan of the same glibc, so even those are bit-exact here."""
    g = golden(name)
    stereo = g.meta.get("stereo", 0)
    if "depth" in g.data:
        out, dm, st = oracle.render(g.scene, g.width, g.height, g.depth, stereo=stereo, depth_map=True)
        assert np.array_equal(dm, g.data["depth"])
        assert (dm > 0).any()
    else:
        out, st = oracle.render(g.scene, g.width, g.height, g.depth, stereo=stereo)
    out, ref = comparable(out, stereo), comparable(g.data["fb"], stereo)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    assert st.rays_ref_equiv == g.meta["rays_total"]


@pytest.mark.parametrize("name", SAMPLED_CASES)
def test_jittered_samples_and_lens_bit_exact(oracle, name):
    """`-n samples` > 1 (ndt.c:505-542): jitter inside the pixel and a lens sample per ray, both from the
    global drand48 stream.  The reference run that made the fixture was single-threaded and recorded
    where the stream stood when the render began (scene programs draw from it too); the oracle starts
    there and walks the pixels in the same order, so it draws the same numbers: same image, same
    trace_kd count."""
    g = golden(name)
    out, st = oracle.render(g.scene, g.width, g.height, g.depth, samples=g.meta["samples"], seed48=g.meta["seed48"],
                            stereo=g.meta.get("stereo", 0))
    assert np.array_equal(out, g.data["fb"]), "max abs diff %g" % np.abs(out - g.data["fb"]).max()
    assert st.rays_ref_equiv == g.meta["rays_total"]


@pytest.mark.parametrize("name", SAMPLED_MODE_CASES)
def test_jittered_samples_in_the_other_modes_bit_exact(oracle, name):
    """-n > 1 as an anaglyph (two adaptive loops per pixel, left eye first), frame-packed, and with a depth map (every sample
    overwrites the pixel's depth: the last one's stays, ndt.c:362-373)."""
    g = golden(name)
    stereo = g.meta.get("stereo", 0)
    if "depth" in g.data:
        out, dm, st = oracle.render(g.scene, g.width, g.height, g.depth, samples=g.meta["samples"], seed48=g.meta["seed48"],
                                    stereo=stereo, depth_map=True)
        assert np.array_equal(dm, g.data["depth"]) and (dm > 0).any()
    else:
        out, st = oracle.render(g.scene, g.width, g.height, g.depth, samples=g.meta["samples"], seed48=g.meta["seed48"], stereo=stereo)
    out, ref = comparable(out, stereo), comparable(g.data["fb"], stereo)
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()
    assert st.rays_ref_equiv == g.meta["rays_total"]
