"""GPU parity of the extract stage (ochip_akaze_batch + the host tail of extract_features) against the
oracle's restatement.  AKAZE's parity with OpenCV is unpinned (no OpenCV here); what is checked is that the
device kernels reproduce the restatement: float32, same operation order, shared host tables, so keypoints
and descriptors are expected to be bit-identical."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _sorted(kp, desc):
    order = np.lexsort((kp[:, 0], kp[:, 1], kp[:, 5]))
    return kp[order], desc[order]


@pytest.mark.parametrize("w,h", [(320, 240), (640, 480), (500, 333), (502, 331), (640, 200), (501, 334)])   # 502: rows not 16-byte aligned; 640 x 200: an octave of 160 x 50 (octaves end below 80 wide or 40 high); odd widths / heights: the general INTER_AREA path between octaves
def test_akaze_matches_restatement_bitwise(ctx, oracle, w, h):
    imgs = np.stack([synth.render_blobs(w, h, seed) for seed in (1, 2, 3)])
    got, (ww, wh) = ctx.akaze_batch(imgs, max_kp=20000)
    assert (ww, wh) == (w, h)
    for i in range(len(imgs)):
        ekp, edesc = oracle.akaze(imgs[i][:, :, 0])     # grey of a grey-valued BGR image is itself
        gkp, gdesc = got[i]                             # same ORDER too: the detection order (level, row, column)
        assert len(gkp) == len(ekp) and len(ekp) > 50   # descriptor windows near the border are excluded (AKAZE)
        assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32))     # positions, sizes, angles, responses
        assert np.array_equal(gdesc, edesc)


def test_grey_and_area_resize(ctx, oracle):
    """Images larger than 1600 px are converted to grey and INTER_AREA-downscaled first (extract_features.cpp:25-27);
    coloured input exercises the fixed-point grey conversion."""
    rng = np.random.default_rng(5)
    w, h = 2000, 1500
    base = synth.render_blobs(w, h, 9)
    tint = rng.integers(0, 40, (h, w, 3), dtype=np.uint8)
    img = np.clip(base.astype(np.int32) + tint - 20, 0, 255).astype(np.uint8)
    got, (ww, wh) = ctx.akaze_batch(img[None], max_kp=30000)
    assert (ww, wh) == (1600, 1200)
    small = oracle.gray_resize(img, ww, wh)
    ekp, edesc = _sorted(*oracle.akaze(small))
    gkp, gdesc = _sorted(*got[0])
    assert len(gkp) == len(ekp) and len(ekp) > 1000
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32))
    assert np.array_equal(gdesc, edesc)


@pytest.mark.parametrize("w,h", [(3200, 2400), (2392, 3200)])
def test_a_scale_of_exactly_two(ctx, oracle, w, h):
    """A 3200-pixel side makes the float scale exactly 0.5, and cv::resize(INTER_AREA) takes its integer path: the vector body
    rounds (a + b + c + d + 2) >> 2 where the general path rounds to even (oracle/akaze.cpp, D5).  Noise makes sums of
    4 n + 2 as common as any; 2392 x 3200 -> 1196 columns: the last 1196 % 16 = 12 are the scalar tail."""
    rng = np.random.default_rng(w)
    base = synth.render_blobs(w, h, 17)[:, :, :1].astype(np.int32)
    img = np.clip(base + rng.integers(0, 4, (h, w, 1)), 0, 255).astype(np.uint8).repeat(3, axis=2)    # (every residue of the sums)
    got, (ww, wh) = ctx.akaze_batch(img[None], max_kp=60000)
    assert (ww, wh) == (w // 2, h // 2)
    small = oracle.gray_resize(img, ww, wh)
    g = img[:, :, 0].astype(np.int32)
    quad = g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2]
    cols = ww - ww % 16
    assert np.array_equal(small[:, :cols], ((quad + 2) >> 2)[:, :cols].astype(np.uint8))     # the oracle does what it says
    assert np.count_nonzero(quad[:, :cols] % 4 == 2) > 1000                                   # ... where it matters
    ekp, edesc = _sorted(*oracle.akaze(small))
    gkp, gdesc = _sorted(*got[0])
    assert len(gkp) == len(ekp)
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)


def test_resize_fallback_for_unaligned_width(ctx, oracle):
    """A source width that is not a multiple of 4 cannot be staged by dword loads: the separate grey and per-tap resize
    kernels run instead and must give the same working image."""
    w, h = 2002, 1500
    img = synth.render_blobs(w, h, 13)
    got, (ww, wh) = ctx.akaze_batch(img[None], max_kp=30000)
    small = oracle.gray_resize(img, ww, wh)
    ekp, edesc = oracle.akaze(small)
    gkp, gdesc = got[0]
    assert (ww, wh) == (1600, 1199) and len(gkp) == len(ekp) > 1000
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)


def test_extract_features_host_tail(ctx, oracle):
    """Whole extract_features: rescale to full resolution, strength order, 8 px NMS, [sparse..., dense...]."""
    imgs = np.stack([synth.render_blobs(800, 600, 11), synth.render_blobs(800, 600, 12)])
    got = host.extract_features_batch(ctx, imgs)
    for i in range(2):
        eloc, est, edesc, ens = oracle.extract_features(imgs[i])
        gloc, gst, gdesc, gns = got[i]
        assert gns == ens and len(gst) == len(est)
        assert np.array_equal(gloc, eloc) and np.array_equal(gst, est) and np.array_equal(gdesc, edesc)
        # test/test_extract_features.cpp:8-75 restated: > 100 features, dense ones near a kept one
        assert len(gst) > 100 and gns < len(gst)
        sparse = gloc[:gns]
        for p in gloc[gns:gns + 50]:
            assert np.min(np.linalg.norm(sparse - p, axis=1)) < 200
    assert host.extract_features_batch(ctx, np.zeros((0, 4, 4, 3), np.uint8)) == []


def test_tied_responses_follow_the_reference_sort(ctx, oracle):
    """A periodic image has keypoints with bit-identical responses.  extract_features orders by response with an
    unstable std::sort (extract_features.cpp:55-56), so the result depends on the order the keypoints arrive in:
    AKAZE's detection order, which is the order the device hands them over in."""
    base = synth.render_blobs(400, 304, 21)
    img = np.ascontiguousarray(np.tile(base, (2, 2, 1)))
    (gloc, gst, gdesc, gns), = host.extract_features_batch(ctx, img[None])
    eloc, est, edesc, ens = oracle.extract_features(img)
    _, counts = np.unique(est, return_counts=True)
    assert int((counts > 1).sum()) > 100         # hundreds of tied groups
    assert gns == ens and np.array_equal(gst, est)
    assert np.array_equal(gloc, eloc) and np.array_equal(gdesc, edesc)


def test_two_views_match(ctx):
    """The extracted descriptors are usable: two renderings of one scene (shift + rotation) match through the
    device matcher with the ratio test, and the matches are geometrically consistent."""
    a = synth.render_blobs(640, 480, 21)
    b = synth.render_blobs(640, 480, 21, shift=(7.3, -4.2), rot=0.15)
    (fa, fb) = host.extract_features_batch(ctx, np.stack([a, b]))
    ctx.descriptors_reserve(2, len(fa[1]) + len(fb[1]))
    ctx.upload_descriptors(0, fa[2])
    ctx.upload_descriptors(1, fb[2])
    raw = ctx.match_batch(np.array([(0, 1)], capi.PAIR_DTYPE), np.array([0], np.uint64), len(fa[1]))
    i1, i2, dist = host.matches_from_device(raw, np.arange(len(fa[1]), dtype=np.uint64), np.arange(len(fb[1]), dtype=np.uint64))
    assert len(i1) > 200
    c, s = np.cos(0.15), np.sin(0.15)
    q = fb[0][i2.astype(int)]
    pred = np.stack([c * (q[:, 0] - 320) - s * (q[:, 1] - 240) + 320 + 7.3, s * (q[:, 0] - 320) + c * (q[:, 1] - 240) + 240 - 4.2], 1)
    err = np.linalg.norm(pred - fa[0][i1.astype(int)], axis=1)
    assert np.mean(err < 3.0) > 0.9


def test_featureless_tiny_and_overflowing_inputs(ctx, oracle):
    """Edge cases of the extract boundary: an image without any structure yields {[], 0} like the reference's empty
    result (extract_features.cpp:20-23 / no keypoints); a tiny image (smaller than one 64 x 32 tile in the upper
    octaves) still matches the restatement; a keypoint budget that is too small fails loudly instead of truncating."""
    flat = np.full((1, 240, 320, 3), 127, np.uint8)
    (loc, st, desc, ns), = host.extract_features_batch(ctx, flat)
    assert len(st) == 0 and ns == 0 and loc.shape == (0, 2) and desc.shape == (0, 8)
    tiny = synth.render_blobs(96, 80, 4)
    got, _ = ctx.akaze_batch(tiny[None], max_kp=5000)
    ekp, edesc = _sorted(*oracle.akaze(tiny[:, :, 0]))
    gkp, gdesc = _sorted(*got[0])
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)
    busy = synth.render_blobs(640, 480, 2)
    with pytest.raises(capi.OchipError, match="max_kp|keypoints|candidates"):
        ctx.akaze_batch(busy[None], max_kp=50)
    # the context is still usable afterwards
    got, _ = ctx.akaze_batch(busy[None], max_kp=20000)
    assert len(got[0][0]) > 500


@pytest.mark.parametrize("w,h", [(1600, 1200), (784, 600), (1584, 1192), (776, 584)])
def test_strip_kernels_on_noisy_views_up_to_the_border(ctx, oracle, w, h):
    """Round 5's register-strip kernels (level_strip_kernel, det_strip_kernel) on views with pixel noise: the diffusion spreads
    whatever the conductivity's reflected taps and the Gaussian's replicated border produce along the image border a few
    pixels further at every level, and descriptors of keypoints near the border sample it.  Widths that are a multiple of 16
    (every level of every octave has an even width: the strips can take them all) but not of the strips' 104 - 120 columns,
    heights that are not a multiple of their rows; 1600 x 1200 is the working size of the bench's 4000 x 3000 views;
    776 x 584 reaches an odd width in the last octave: the tile kernels everywhere, whatever the route."""
    rng = np.random.default_rng(w)
    base = synth.render_blobs(w, h, 31)
    img = np.clip(base.astype(np.int32) + rng.integers(0, 40, (h, w, 1)) - 20, 0, 255).astype(np.uint8)
    got, (ww, wh) = ctx.akaze_batch(img[None], max_kp=40000)
    assert (ww, wh) == (w, h)
    ekp, edesc = oracle.akaze(img[:, :, 0])
    gkp, gdesc = got[0]
    assert len(gkp) == len(ekp) > 500
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)


def test_the_mixed_route_of_a_bench_chunk(oracle, monkeypatch):
    """The default routing of a bench chunk - strips on the levels of >= 32 Mpixel per launch (octaves 0 - 1 of 100 views), tiles
    and the tile form of the contrast pass on the rest - on a batch small enough for the restatement: OCHIP_STRIP_MIN_PIXELS
    puts the threshold between the second and the third octave of three 784 x 600 views."""
    monkeypatch.setenv("OCHIP_STRIP_MIN_PIXELS", str(3 * 392 * 300))      # octave 1 and up are strips, octaves 2 - 3 tiles
    c = capi.Context(0)
    rng = np.random.default_rng(5)
    imgs = []
    for k in range(3):
        base = synth.render_blobs(784, 600, 40 + k)
        imgs.append(np.clip(base.astype(np.int32) + rng.integers(0, 30, (600, 784, 1)) - 15, 0, 255).astype(np.uint8))
    got, (ww, wh) = c.akaze_batch(np.stack(imgs), max_kp=40000)
    assert (ww, wh) == (784, 600)
    for k in range(3):
        ekp, edesc = oracle.akaze(imgs[k][:, :, 0])
        gkp, gdesc = got[k]
        assert len(gkp) == len(ekp) > 300
        assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)
    c.close()


@pytest.mark.parametrize("hooks", ["tile_levels,tile_det", "strip_levels,strip_det"])
def test_tile_and_strip_kernels_agree(hooks):
    """The scale space and the detector have two forms: the register-strip kernels of round 5 (level_strip_kernel,
    det_strip_kernel) for launches with enough strips to fill the device (>= 32 Mpixel per level and launch: the first two
    octaves of a 100-image chunk), and the tile kernels of rounds 2 - 4 (blur_fused / nld_fused / det_maxima) for the rest
    and for levels of odd width.  A single test image takes the tiles everywhere; OCHIP_TEST_HOOKS=strip_levels,strip_det
    sends every level the strips can take through them, tile_levels,tile_det every level through the tiles.  This file's
    parity tests run again in a child per route."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, OCHIP_TEST_HOOKS=hooks)
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                        "restatement_bitwise or grey_and_area or host_tail or tied_responses or noisy_views"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("hooks", ["sup_generic", "sup_small_lists", "sup_generic,sup_small_lists"])
def test_suppression_routes_agree(hooks):
    """Find_Scale_Space_Extrema's suppression (csrc/akaze.hip, suppress_*_kernel) has code the default scale space never
    reaches on the test images: mask probes written for any radius (the branch-free ones cover the radii of sigma_size 2 - 4)
    and the part of a level's waiting list that does not fit its 2 048 LDS entries.  OCHIP_TEST_HOOKS=sup_generic sends every
    probe through the former, sup_small_lists keeps 64 entries in LDS; this file's parity tests run again in a child per
    route."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, OCHIP_TEST_HOOKS=hooks)
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                        "restatement_bitwise or noisy_views or mixed_route"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("kind", ["periodic", "noise", "lattice"])
def test_order_dependent_suppression_on_hard_images(ctx, oracle, kind):
    """Find_Scale_Space_Extrema's passes (OpenCV 4.x) decide by the ORDER of the turns and by strict comparisons of responses;
    the device takes the turns in dependency rounds.  Images that stress exactly that: a periodic texture (many maxima with
    EQUAL responses next to each other - the earlier one stays), pure noise (maxima at the highest density the 3 x 3 test allows:
    the longest chains of dependent turns, waiting lists beyond their LDS part) and a sine lattice (maxima lined up along rows, the
    raster order's worst case).  Bit-identical keypoints and descriptors, as everywhere."""
    w, h = 640, 480
    rng = np.random.default_rng(7)
    if kind == "periodic":
        cell = synth.render_blobs(64, 48, 3)[:, :, 0]
        img = np.tile(cell, (h // 48, w // 64))
    elif kind == "noise":
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    else:
        x = np.arange(w)[None, :]
        y = np.arange(h)[:, None]
        img = (127 + 100 * np.sin(x * 0.5) * np.sin(y * 0.37)).astype(np.uint8)
    bgr = np.ascontiguousarray(np.repeat(img[:, :, None], 3, axis=2))
    got, _ = ctx.akaze_batch(bgr[None], max_kp=60000)
    ekp, edesc = oracle.akaze(img)
    gkp, gdesc = got[0]
    assert len(gkp) == len(ekp) > 100, (len(gkp), len(ekp))
    assert np.array_equal(gkp.view(np.uint32), ekp.view(np.uint32)) and np.array_equal(gdesc, edesc)


def test_verbose_extraction_reports_the_suppression_rounds(capfd, monkeypatch, oracle):
    """OCHIP_VERBOSE=extract makes the suppression keep rounds / time / waiting points per (pass, image, level) and print their
    maxima (profiles/r06_suppression_levels.txt comes from it): the diagnostic path must not change the result."""
    monkeypatch.setenv("OCHIP_VERBOSE", "extract")
    img = synth.render_blobs(640, 480, 11)
    c = capi.Context(0)
    got, _ = c.akaze_batch(img[None], max_kp=20000)
    c.close()
    err = capfd.readouterr().err
    assert err.count("suppression pass") == 3 and "rounds/time/points" in err
    ekp, edesc = oracle.akaze(img[:, :, 0])
    assert np.array_equal(got[0][0].view(np.uint32), ekp.view(np.uint32)) and np.array_equal(got[0][1], edesc)
