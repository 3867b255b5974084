"""GPU-less logic tests: the UNMODIFIED product sources (kernels + C ABI) compiled against the fiber-based
HIP emulation in tests/emu and checked bit for bit against the oracle.  These do not replace the -m gpu parity
tests; they keep indexing/logic regressions out of the GPU runs."""
import os

import numpy as np
import pytest

import emu_util
import parity
from helpers import GOLDEN, load_glue_golden, synth_grey
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params, I2sError
from oracle import pipeline as opipe


@pytest.fixture(scope="module")
def lib():
    return emu_util.emu_library()


def test_small_synthetic_with_internals(lib):
    det = Detector(0, 2, 300, 260, lib=lib)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in (0, 1)]
    dets = parity.run_and_compare(det, imgs, internals=True)
    for s, d in zip((0, 1), dets):
        occ = synth.occupancy(s, 9, 8)
        assert d.board_ready and (d.full_board[:9, :8] == occ).all()
    det.close()


def test_small_pass_then_full_pass_on_one_context(lib):
    """Round 4: per-context buffers sized on first use (the packed full records' offsets) must be sized for the CONTEXT's batch, not
    for the pass that happens to come first: one image, then three, full records both times."""
    det = Detector(0, 3, 320, 300, lib=lib)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in (5, 6, 7)]
    parity.run_and_compare(det, imgs[:1])
    parity.run_and_compare(det, imgs)
    det.close()


def test_multi_pass_ragged_batch(lib):
    """3 images of different sizes through a context holding 2 per device pass."""
    det = Detector(0, 2, 320, 300, lib=lib)
    a = synth.synth_diagram(2, geom=synth.GEOM_SMALL)[0]
    b = np.ascontiguousarray(a[:200, :250])
    c = np.pad(a, ((10, 30), (5, 40)), constant_values=255)
    parity.run_and_compare(det, [a, b, c])
    det.close()


@pytest.mark.parametrize("name", ["no_circles.jpg", "ex9.jpg"])
def test_reference_fixture(lib, name):
    img = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", name))
    det = Detector(0, 1, img.shape[1], img.shape[0], lib=lib)
    parity.run_and_compare(det, [img], internals=True)
    det.close()


def test_mixed_colour_batch_and_unfused_canny(lib):
    """A greyscale and a colour source in one device pass (the fused grey-plane Canny handles the first, the colour kernel
    the second), then the same batch with a main-Canny low threshold that differs from HoughCircles' (separate passes)."""
    a = synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0]
    rng = np.random.default_rng(11)
    col = np.stack([a, np.clip(a.astype(int) + rng.integers(-40, 40, a.shape), 0, 255).astype(np.uint8), a[::-1]], axis=-1)
    det = Detector(0, 2, a.shape[1], a.shape[0], lib=lib)
    parity.run_and_compare(det, [a, np.ascontiguousarray(col)], internals=True)
    parity.run_and_compare(det, [a, np.ascontiguousarray(col)], params=Params(canny_lo=40, canny_hi=150),
                           oracle_kwargs=dict(canny=(40, 150)))
    det.close()


def test_noisy_small_synthetic(lib):
    det = Detector(0, 1, 300, 260, lib=lib)
    parity.run_and_compare(det, [synth.synth_diagram(4, noisy=True, geom=synth.GEOM_SMALL)[0]], internals=True)
    det.close()


def test_tiny_images(lib):
    det = Detector(0, 4, 70, 70, lib=lib)
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for (h, w) in [(1, 1), (2, 5), (7, 3), (33, 65)]]
    parity.run_and_compare(det, imgs, internals=True)
    det.close()


def test_reclassify(lib):
    """apply_black_thresh (img2sgf.py:762-766): identify_board only, with a new threshold / alignment."""
    det = Detector(0, 1, 300, 260, lib=lib)
    img = synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0]
    det.detect_batch([img])
    d2 = det.classify(0, 1, Params(black_threshold=250, alignment=(3, 1)))[0]
    ref = opipe.process_image(img, black_thr=250, alignment=(3, 1))
    parity.compare_detection(d2, ref)
    assert d2.num_white_stones == 0
    det.close()


G = load_glue_golden()


def test_glue_golden_through_c_abi(lib):
    """find_grid() on injected rho lists / circles: the device glue against vectors produced by the reference."""
    det = Detector(0, 1, 1000, 1000, lib=lib)
    for entry in G["cases"]:
        case, exp = entry["case"], entry["expect"]
        grey = synth_grey(case["w"], case["h"], case["seed"])
        d = det.grid_from_lines(grey, np.array(case["circles"], np.float32).reshape(-1, 3), case["hlines"], case["vlines"],
                                Params(line_threshold=case["threshold"], black_threshold=case["black_thr"],
                                       alignment=case["alignment"]))
        name = case["name"]
        assert d.found_grid == exp["found_grid"] and d.valid_grid == exp["valid_grid"], name
        assert d.board_ready == exp["board_ready"], name
        np.testing.assert_array_equal(d.hcentres, np.array(exp["hcentres"]), err_msg=name)
        np.testing.assert_array_equal(d.vcentres, np.array(exp["vcentres"]), err_msg=name)
        assert (d.hsize, d.vsize) == (exp["hsize"], exp["vsize"]), name
        if exp["valid_grid"]:
            np.testing.assert_array_equal(d.hcentres_complete, np.array(exp["hcentres_complete"]), err_msg=name)
            np.testing.assert_array_equal(d.vcentres_complete, np.array(exp["vcentres_complete"]), err_msg=name)
            assert (d.hspace, d.vspace) == (exp["hspace"], exp["vspace"]), name
            np.testing.assert_array_equal(d.circles, np.array(exp["kept_circles"], np.float32).reshape(-1, 3), err_msg=name)
        if exp["board_ready"]:
            np.testing.assert_array_equal(d.detected_board, np.array(exp["detected_board"]), err_msg=name)
            np.testing.assert_array_equal(d.full_board, np.array(exp["full_board"]), err_msg=name)
            np.testing.assert_array_equal(d.stone_brightnesses, np.array(exp["stone_brightnesses"]), err_msg=name)
            assert (d.num_black_stones, d.num_white_stones, d.side_to_move) == (
                exp["num_black_stones"], exp["num_white_stones"], exp["side_to_move"]), name
            assert d.sgf == exp["sgf"], name
    det.close()


def test_errors(lib):
    det = Detector(0, 1, 64, 64, lib=lib)
    with pytest.raises(I2sError):
        det.detect_batch([np.zeros((65, 10), np.uint8)])
    with pytest.raises(I2sError):
        det.detect_batch([np.zeros((10, 10), np.uint8)], Params(hc_max_radius=40))
    det.close()


def test_device_contrast_brightness_matches_pillow(lib):
    """ImageEnhance.Contrast / .Brightness (img2sgf.py:141-149) on the device, from the raw decoded RGB."""
    from img2sgf_amd import preprocess
    for name, (c, b) in [("no_circles.jpg", (70, 50)), ("ex9.jpg", (70, 50)), ("ex9.jpg", (25, 85))]:
        path = os.path.join(GOLDEN, "test_images", name)
        raw = np.array(preprocess.load_image(path))
        want = preprocess.enhance(preprocess.load_image(path), c, b)
        det = Detector(0, 1, raw.shape[1], raw.shape[0], lib=lib)
        d = det.detect_batch([raw], Params(contrast=c, brightness=b))[0]
        np.testing.assert_array_equal(det.fetch_source(0), want)
        parity.compare_detection(d, opipe.process_image(want))
        det.close()


def test_device_rotate_crop_matches_pillow(lib):
    """crop_and_rotate_image (img2sgf.py:110-114) on the device against Pillow itself: the staged source must equal
    Image.rotate(NEAREST, fillcolor white, center).crop(box) bit for bit, for L and RGB images, boxes reaching outside the
    image included; the detection must equal the oracle's on the Pillow-transformed image."""
    from PIL import Image
    from img2sgf_amd import preprocess
    rng = np.random.default_rng(21)
    base = synth.synth_diagram(5, geom=synth.GEOM_SMALL)[0]
    h, w = base.shape
    rgb = np.ascontiguousarray(np.stack([base, base[::-1], rng.integers(0, 256, base.shape, dtype=np.uint8)], axis=-1))
    cases = [(base, 0.0, None), (base, 3.5, (10, 12, w - 20, h - 8)), (rgb, -17.25, (5, 0, w - 1, h - 30)),
             (base, 90.0, (0, 0, w, h)), (rgb, 180.0, None), (base, 45.0, (-15, -10, w + 12, h + 9)),
             (rgb, 0.0, (30, 40, 150, 160)), (base, 271.3, (20, 20, 200, 180))]
    det = Detector(0, 3, w + 40, h + 40, lib=lib)
    imgs = [c[0] for c in cases]
    xfs = [preprocess.xform((w, h), ang, sel) for (_, ang, sel) in cases]
    dets = det.detect_batch(imgs, Params(), xforms=xfs)
    n_last = (len(cases) - 1) % 3 + 1
    for k, ((img, ang, sel), d) in enumerate(zip(cases, dets)):
        s = sel if sel is not None else (0, 0, w, h)
        ref = np.array(Image.fromarray(img).rotate(angle=-ang, fillcolor="white", center=preprocess.rectangle_centre(s)).crop(s))
        if k >= len(cases) - n_last:
            got = det.fetch_source(k - (len(cases) - n_last), 1 if img.ndim == 2 else 3)
            assert got.shape == ref.shape and (got == ref).all(), (k, ang, sel)
        parity.compare_detection(d, opipe.process_image(ref))
    det.close()


def test_scheduled_ragged_batch_returns_input_order(lib):
    """Params.schedule: passes formed over the images sorted by area; results must come back in input order and equal the
    unscheduled ones (with and without a device pre-transform)."""
    import ctypes as C
    from img2sgf_amd import preprocess
    a = synth.synth_diagram(6, geom=synth.GEOM_SMALL)[0]
    imgs = [np.ascontiguousarray(a[:200, :220]), np.ascontiguousarray(a[:120, :130]), np.pad(a[:150, :160], ((4, 9), (7, 3)), constant_values=255),
            np.ascontiguousarray(a[20:170, 10:160]), np.ascontiguousarray(a[:60, :70])]
    det = Detector(0, 3, 330, 300, lib=lib)
    plain = det.detect_batch(imgs, Params(), full=False)
    sched = det.detect_batch(imgs, Params(schedule=True), full=False)
    for k in range(len(imgs)):
        assert bytes(plain[k]) == bytes(sched[k]), k
    # the resident ("last") pass of a scheduled call holds the largest images, in area order
    order = sorted(range(len(imgs)), key=lambda i: imgs[i].shape[0] * imgs[i].shape[1])
    n_last = (len(imgs) - 1) % 3 + 1
    for j, i in enumerate(order[len(imgs) - n_last:]):
        np.testing.assert_array_equal(det.fetch_plane(j, "grey"), imgs[i])
    xfs = [preprocess.xform((i.shape[1], i.shape[0]), 2.0 * k, None) for k, i in enumerate(imgs)]
    plain = det.detect_batch(imgs, Params(), full=True, xforms=xfs)
    sched = det.detect_batch(imgs, Params(schedule=True), full=True, xforms=xfs)
    for k in range(len(imgs)):
        assert plain[k].sgf == sched[k].sgf and plain[k].status == sched[k].status, k
        np.testing.assert_array_equal(plain[k].circles_all, sched[k].circles_all)
    det.close()


def test_jpeg_decode_matches_pillow(lib):
    """SURVEY 8f-4 under the emulator: reference fixtures and Pillow-encoded images (4:4:4 / 4:2:2 / 4:2:0 / grey, restart
    intervals, progressive) decoded by jpeg_host.h + k_jpeg.h must equal Image.open(...).convert("RGB"); CMYK is refused."""
    import io
    from PIL import Image
    blobs = []
    for n in ("no_circles.jpg", "ex9.jpg"):
        with open(os.path.join(GOLDEN, "test_images", n), "rb") as f:
            blobs.append(f.read())
    rng = np.random.default_rng(31)
    base = synth.synth_diagram(8, geom=synth.GEOM_SMALL)[0]
    col = np.stack([base, np.roll(base, 5, 0), rng.integers(0, 256, base.shape, dtype=np.uint8)], -1)
    for kw in (dict(subsampling=0, quality=90), dict(subsampling=1, quality=60, restart_marker_blocks=5),
               dict(subsampling=2, quality=35, optimize=True, restart_marker_rows=1),
               dict(subsampling=2, quality=50, progressive=True), dict(subsampling=0, quality=20, progressive=True, restart_marker_blocks=7)):
        buf = io.BytesIO()
        Image.fromarray(col[:93, :131]).save(buf, "JPEG", **kw)
        blobs.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(base[:67, :91]).save(buf, "JPEG", quality=75)
    blobs.append(buf.getvalue())
    # the same coefficients as one scan per component (tests/jpeg_transcode.py): a sequential file Pillow reads but cannot write
    import jpeg_transcode
    buf = io.BytesIO()
    Image.fromarray(col[:61, :83]).save(buf, "JPEG", quality=70, subsampling=2)
    blobs.append(jpeg_transcode.to_non_interleaved(buf.getvalue(), [1, 2, 0]))
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs]
    det = Detector(0, len(blobs), 300, 260, lib=lib)
    dets = det.detect_jpeg(blobs, Params(jpeg_entropy_device=2), full=True)
    for k, (d, r) in enumerate(zip(dets, refs)):
        np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="jpeg %d" % k)
        parity.compare_detection(d, opipe.process_image(r))
    # Huffman decoding on host threads (mode 0) and the default split (sequential files on the device, progressive on the host)
    for mode in (0, 1):
        det.detect_jpeg(blobs, Params(jpeg_entropy_device=mode), full=False)
        for k, r in enumerate(refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="jpeg %d, entropy mode %d" % (k, mode))
        assert (det.jpeg_last_rounds() > 0) == (mode == 1)         # the parallel decoder ran (and needed more than the first round)
    # first passes that are not disjoint (repeated bands, runs past the band's end: tests/jpeg_transcode.py::to_progressive): the
    # device's share ends at the clash / the file goes to the serial decoder; Pillow's pixels on every path
    buf = io.BytesIO()
    Image.fromarray(col[:40, :56]).save(buf, "JPEG", quality=80, subsampling=2)
    clean = [(0, 1, 5, 1), (1, 1, 63, 1), (2, 1, 63, 1), (0, 6, 63, 1)]
    odd = [jpeg_transcode.to_progressive(buf.getvalue(), clean),
           jpeg_transcode.to_progressive(buf.getvalue(), [(0, 1, 5, 1), (0, 3, 9, -1), (1, 1, 63, 1), (2, 1, 63, 1), (0, 6, 63, 1)]),
           jpeg_transcode.to_progressive(buf.getvalue(), clean, overrun=(0, 3))]
    odd_refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in odd]
    for mode in (0, 1, 2):
        det.detect_jpeg(odd, Params(jpeg_entropy_device=mode), full=False)
        for k, r in enumerate(odd_refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="non-conforming progressive file %d, entropy mode %d" % (k, mode))
    # files still iterating at the limit are handed to the serial decoder
    det.jpeg_set_max_rounds(1)
    det.detect_jpeg(blobs, Params(), full=False)
    assert det.jpeg_last_rounds() == 1 and det.jpeg_last_handed_back() > 0
    for k, r in enumerate(refs):
        np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="jpeg %d, one round allowed" % k)
    det.jpeg_set_max_rounds(2048)
    with pytest.raises(I2sError):
        det.detect_jpeg([blobs[0][:len(blobs[0]) // 2] + b"\xff\xd9"], Params(jpeg_entropy_device=2), full=False)      # truncated entropy data, device path
    buf = io.BytesIO()
    Image.fromarray(np.zeros((24, 24, 4), np.uint8), "CMYK").save(buf, "JPEG")
    with pytest.raises(I2sError):
        det.detect_jpeg([buf.getvalue()], Params(), full=False)
    det.close()


def test_canny_byte_walk_emulated(lib):
    """Round 4: Sobel + suppression of two-valued bands in bytes, restarts in 16-bit pairs -- the emulated twin of
    tests/test_gpu_parity.py::test_canny_byte_walk_shapes_thresholds_and_restarts (small shapes, the thresholds around the multiples of
    255 the byte walk compares in, a stray grey pixel in the middle of a band and in the apron of the next)."""
    from oracle import cv_oracle as cvo
    rng = np.random.default_rng(405)

    def check(det, imgs, prm):
        det.detect_batch(imgs, prm, full=False)
        hc_lo = max(1, prm.hc_param1 // 2)
        for i, im in enumerate(imgs):
            np.testing.assert_array_equal(det.fetch_plane(i, "edges"), cvo.canny(im, prm.canny_lo, prm.canny_hi), err_msg="edges %s %s" % (im.shape, prm))
            for v, name in enumerate(parity.VARIANT_PLANES):
                want = cvo.canny(det.fetch_plane(i, name), hc_lo, prm.hc_param1)
                got = (det.fetch_plane(i, 9 + 1 + v) == 2).astype(np.uint8) * 255
                np.testing.assert_array_equal(got, want, err_msg="%s %s %s" % (name, im.shape, prm))

    det = Detector(0, 2, 264, 70, lib=lib)
    det.set_debug(True)
    shapes = [(1, 1), (3, 5), (9, 4), (34, 257), (66, 13), (40, 260)]
    imgs = []
    for k, (h, w) in enumerate(shapes):
        im = np.where(rng.random((h, w)) < (0.5, 0.15)[k % 2], 0, 255).astype(np.uint8)
        if h > 20:
            im[h // 3:h // 3 + 3, :] = 0; im[:, w // 2:w // 2 + 2] = 255
        imgs.append(im)
    for prm in (Params(), Params(canny_lo=254, canny_hi=255, hc_param1=510), Params(canny_lo=-5, canny_hi=0, hc_param1=1),
                Params(canny_lo=255, canny_hi=1275, hc_param1=1021)):
        for k in range(0, len(imgs), 2):
            check(det, imgs[k:k + 2], prm)
    base = imgs[3]
    a = base.copy(); a[17, 255] = 77
    b = base.copy(); b[32, 3] = 254
    check(det, [a, b], Params())
    det.close()


def test_blur_bank_shapes_emulated(lib):
    """k_blur's border machinery (byte permutations of the lane's dword triple, reflect / replicate rows, the 7-row ring)
    and the bit-plane medians on awkward shapes -- the emulated twin of tests/test_gpu_parity.py::test_blur_bank_shapes."""
    from oracle import cv_oracle as cvo
    rng = np.random.default_rng(78)
    shapes = [(1, 1), (2, 3), (3, 2), (5, 7), (9, 4), (7, 8), (8, 13), (66, 12), (13, 257), (70, 260)]
    det = Detector(0, 2, 260, 70, lib=lib)
    for k in range(0, len(shapes), 2):
        imgs = []
        for n, (h, w) in enumerate(shapes[k:k + 2]):
            im = rng.integers(0, 256, (h, w), dtype=np.uint8)
            if (k + n) % 3 == 0:
                im = np.where(im < 128, 0, 255).astype(np.uint8)
            imgs.append(im)
        det.detect_batch(imgs, full=False)
        for i, im in enumerate(imgs):
            for name, kk in (("gauss3", 3), ("gauss5", 5), ("gauss7", 7)):
                np.testing.assert_array_equal(det.fetch_plane(i, name), cvo.gaussian_blur(im, kk, kk), err_msg="%s %s" % (name, im.shape))
            for name, kk in (("median3", 3), ("median5", 5), ("median7", 7)):
                np.testing.assert_array_equal(det.fetch_plane(i, name), cvo.median_blur(im, kk), err_msg="%s %s" % (name, im.shape))
            np.testing.assert_array_equal(det.fetch_plane(i, "grey"), im)
    # few grey levels: bit planes that repeat their upper neighbour are skipped by the bit-serial medians (med_repeat)
    levels = [(0, 255), (0, 128, 255), (0, 64, 192, 255), (17, 17), (0xF0, 0x0F, 0xFF, 0x00), (1, 2, 3)]
    for lv in levels:
        im = np.array(lv, np.uint8)[rng.integers(0, len(lv), (41, 60))]
        det.detect_batch([im], full=False)
        for name, kk in (("median3", 3), ("median5", 5), ("median7", 7)):
            np.testing.assert_array_equal(det.fetch_plane(0, name), cvo.median_blur(im, kk), err_msg="%s levels %s" % (name, lv))
    im = (rng.integers(0, 256, (41, 60)) & 0xF0).astype(np.uint8)
    det.detect_batch([im], full=False)
    np.testing.assert_array_equal(det.fetch_plane(0, "median7"), cvo.median_blur(im, 7))
    # k_blur's two-valued speculation: a 0 / 255 image stays in that mode, one stray grey pixel restarts its band in the general one
    base = np.where(rng.integers(0, 2, (70, 260)) == 0, 0, 255).astype(np.uint8)
    for spot in (None, (0, 0, 128), (63, 255, 254), (64, 256, 1), (66, 259, 100), (69, 3, 77)):
        im = base.copy()
        if spot: im[spot[0], spot[1]] = spot[2]
        det.detect_batch([im], full=False)
        for name, kk in (("gauss3", 3), ("gauss5", 5), ("gauss7", 7)):
            np.testing.assert_array_equal(det.fetch_plane(0, name), cvo.gaussian_blur(im, kk, kk), err_msg="%s spot %s" % (name, spot))
        for name, kk in (("median3", 3), ("median5", 5), ("median7", 7)):
            np.testing.assert_array_equal(det.fetch_plane(0, name), cvo.median_blur(im, kk), err_msg="%s spot %s" % (name, spot))
    det.detect_batch([np.full((20, 40), 255, np.uint8)], Params(gauss_kernel_mode=1), full=False)     # tap sums != 256: integer kernels
    np.testing.assert_array_equal(det.fetch_plane(0, "gauss7"), cvo.gaussian_blur(np.full((20, 40), 255, np.uint8), 7, 7, 1))
    det.close()


def test_hysteresis_chain_through_many_tiles(lib):
    """A weak edge that snakes through many 64 x 32 tiles, anchored by one strong seed: more passes than the plain launches in
    front (one, on a fresh context) -- the rest runs inside the persistent tail kernel, which the emulated build launches as a
    single workgroup; a second call on the same context then starts with as many plain launches as the first one needed."""
    h, w = 160, 256
    img = np.full((h, w), 100, np.uint8)
    for k, y in enumerate(range(20, h - 20, 24)):
        img[y:y + 12, 10:w - 10] = 130
    for k, y in enumerate(range(20, h - 44, 24)):
        x = w - 30 if k % 2 == 0 else 10
        img[y:y + 36, x:x + 20] = 130
    img[20:32, 10:14] = 255
    det = Detector(0, 1, w, h, lib=lib)
    parity.run_and_compare(det, [img], internals=False)
    parity.run_and_compare(det, [img], internals=False)
    det.close()


def test_validate_grid_from_none_to_1024_centres_emulated(lib):
    det = Detector(0, 1, 600, 600, lib=lib)
    parity.check_validate_grid_capacity(det, 60)
    det.close()


def test_find_lines_on_degenerate_images_emulated(lib):
    det = Detector(0, 1, 140, 140, lib=lib)
    parity.check_find_lines_degenerate(det, 140)
    det.close()


def test_call_sequence_on_one_context_emulated(lib):
    """The emulated twin of tests/test_gpu_fuzz_sequences.py (state that leaks from one call into the next lives in the host code,
    which the emulated build shares): one context, six calls of 1 .. 4 images of at most 88 pixels a side."""
    from test_gpu_fuzz_sequences import run_call_sequence
    rng = np.random.default_rng(70001)
    det = Detector(0, 2, 88, 88, lib=lib)
    run_call_sequence(det, rng, "emu", n_calls=6, side=88, max_images=4)
    det.close()


@pytest.mark.parametrize("seed", [1, 4, 7, 12])
def test_extreme_parameters_emulated(lib, seed):
    """Emulated twin of tests/test_gpu_fuzz_extreme.py: the edges of the parameter envelope on images of at most 64 x 44 pixels."""
    from test_gpu_fuzz_extreme import run_extreme_seed
    run_extreme_seed(lambda nb, w, h: Detector(0, nb, w, h, lib=lib), seed, side=64, n_images=2)


def test_capacity_by_lines_first_and_last_in_full_batch(lib):
    """Emulated twin of the GPU test of the same name (packed full records sized from the board record's circle
    count and written from the result record's)."""
    det = Detector(0, 4, 3000, 300, lib=lib)
    parity.check_capacity_by_lines_in_full_batch(det, [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in (0, 1)])
    det.close()


@pytest.mark.parametrize("seed", [0, 2, 5])
def test_fuzz_seed_emulated(lib, seed):
    """Emulated twin of tests/test_gpu_fuzz.py::test_fuzz_against_oracle: the same seeds' images and parameters (one default set, one
    non-default switch set, one random parameter set) on the emulated kernels.  tests/stress/emulated_fuzz.py runs hundreds."""
    from test_gpu_fuzz import run_fuzz_seed
    run_fuzz_seed(lambda nb, w, h: Detector(0, nb, w, h, lib=lib), seed)


def test_preprocessing_fuzz_seed_emulated(lib):
    from test_gpu_fuzz import run_preprocessing_fuzz_seed
    run_preprocessing_fuzz_seed(lambda nb, w, h: Detector(0, nb, w, h, lib=lib), 1)


def test_jpeg_handback_beside_redo_emulated(lib):
    """Emulated twin of tests/test_gpu_jpeg.py::test_handback_beside_redo_in_one_pass (the skip mask of k_je_scan / k_je_write)."""
    from test_gpu_jpeg import run_handback_beside_redo
    run_handback_beside_redo(lambda nb, w, h: Detector(0, nb, w, h, lib=lib))
