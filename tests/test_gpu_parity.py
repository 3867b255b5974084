"""Parity tests proper: the HIP path on a real MI355X, through the C ABI, against the oracle (bit-exact: all
integer/byte/index work; the few float32 steps - Hough steps, radius bins - are IEEE-exact operations and must
also match exactly), the committed golden vectors, and size-independent properties at the benchmark size."""
import os

import numpy as np
import pytest

import parity
import switches
from helpers import GOLDEN, free_port, load_glue_golden, synth_grey
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params, board_to_sgf
from oracle import cv_oracle as cvo
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu

EX1_SGF = "(;GM[1]FF[4]SZ[19]\nPL[W]\nAW[cn][jq][nq][qf][qj]\nAB[co][dd][dp][fp][nd][pd][pn][pp][ql]\n)\n"
IMAGES = ["ex%d.jpg" % i for i in range(1, 18)] + ["no_circles.jpg"]


def test_native_library_loaded():
    """The suite ran on an MI355X, on the product library: the context's device says gfx950 (the CPU emulation of tests/emu, which
    tools/gpu_suite_on_emulator.sh puts under the product's file name, says "emulated" and fails here), and the file is
    img2sgf_amd/libi2s_hip.so itself unless I2S_EXPERIMENT=1 declares an experiment build (tools/experiments/ab.sh)."""
    from img2sgf_amd import _lib
    lib = _lib.load()
    assert lib.path.endswith("libi2s_hip.so")
    if os.environ.get("I2S_EXPERIMENT") != "1":
        assert os.path.realpath(lib.path) == os.path.realpath(_lib.LIB_PATH), "I2S_LIBRARY redirects the suite to %s" % lib.path
    with open("/proc/self/maps") as f:
        assert os.path.realpath(lib.path) in f.read()
    det = Detector(0, 1, 64, 64)
    arch = det.arch
    det.close()
    assert arch.startswith("gfx950"), "the library's context runs on %r, not on an MI355X" % arch


def test_small_synthetic_with_internals():
    det = Detector(0, 2, 300, 260)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(4)]
    parity.run_and_compare(det, imgs, internals=True)
    det.close()


def test_mixed_colour_batch_and_unfused_canny():
    """Greyscale + colour sources in one device pass (fused grey-plane Canny / colour Canny side by side), then a main-Canny
    low threshold that differs from HoughCircles' (the two Cannys of the grey plane run as separate passes)."""
    a = synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0]
    rng = np.random.default_rng(11)
    col = np.ascontiguousarray(np.stack(
        [a, np.clip(a.astype(int) + rng.integers(-40, 40, a.shape), 0, 255).astype(np.uint8), a[::-1]], axis=-1))
    det = Detector(0, 2, a.shape[1], a.shape[0])
    parity.run_and_compare(det, [a, col], internals=True)
    parity.run_and_compare(det, [a, col], params=Params(canny_lo=40, canny_hi=150), oracle_kwargs=dict(canny=(40, 150)))
    parity.run_and_compare(det, [col, a, col], params=Params(canny_lo=50, canny_hi=120), oracle_kwargs=dict(canny=(50, 120)))
    det.close()


def test_tiny_and_ragged():
    det = Detector(0, 8, 330, 300)      # one device pass, so every image's planes are still resident for the comparison
    rng = np.random.default_rng(5)
    a = synth.synth_diagram(2, geom=synth.GEOM_SMALL)[0]
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for (h, w) in [(1, 1), (2, 5), (7, 3), (33, 65), (130, 129)]]
    imgs += [a, np.ascontiguousarray(a[:200, :250]), np.pad(a, ((10, 30), (5, 40)), constant_values=255)]
    parity.run_and_compare(det, imgs, internals=True)
    det.close()
    det = Detector(0, 3, 330, 300)      # and the same images through three passes
    parity.run_and_compare(det, imgs)
    det.close()


def test_synthetic_1024_with_internals():
    """BASELINE configs[1]: single 1024x1024 synthetic diagram, every stage against the oracle."""
    det = Detector(0, 1, 1024, 1024)
    img, occ = synth.synth_diagram(0)
    d = parity.run_and_compare(det, [img], internals=True)[0]
    assert d.board_ready and d.threshold == 96 and (d.full_board == occ).all()
    det.close()


def test_noisy_synthetic_1024_with_internals():
    """BASELINE configs[1] "noisy" variant (N(0, 6^2) noise): many more weak edges, long hysteresis chains, dense vote
    accumulators -- every plane, accumulator, circle and line must still match the oracle."""
    det = Detector(0, 2, 1024, 1024)
    imgs = [synth.synth_diagram(s, noisy=True)[0] for s in (0, 1)]
    parity.run_and_compare(det, imgs, internals=True)
    det.close()


def test_synthetic_1024_batch_multi_pass():
    det = Detector(0, 4, 1024, 1024)
    imgs, occs = synth.synth_batch(range(100, 110))
    dets = parity.run_and_compare(det, list(imgs))
    for d, occ in zip(dets, occs):
        assert (d.full_board == occ).all()
    det.close()


@pytest.mark.parametrize("sw", switches.NAMES)
@pytest.mark.parametrize("name", IMAGES)
def test_reference_image(name, sw):
    """BASELINE configs[4]: the reference's 18 fixtures, Pillow pre-processing on the host, every plane / list / record and the
    SGF bytes against the oracle -- under every OpenCV-version switch set (tests/switches.py): the grey weights of
    3.x on the colour fixtures, the plain-rounded Gaussian taps (integer kernels), both HoughLines angle counts."""
    img = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", name))
    det = Detector(0, 1, img.shape[1], img.shape[0])
    d = parity.run_and_compare(det, [img], params=switches.params(sw), oracle_kwargs=dict(compat=switches.compat(sw)))[0]
    if name == "ex1.jpg":
        assert d.sgf == EX1_SGF      # the reference's only recorded result (screenshot.jpg), the same under every switch set
    det.close()


@pytest.mark.parametrize("sw", switches.NAMES)
def test_all_reference_images_one_mixed_batch(sw):
    imgs = [opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", n)) for n in IMAGES]
    det = Detector(0, 6, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    boards = det.detect_batch(imgs, switches.params(sw), full=False)
    for n, img, b in zip(IMAGES, imgs, boards):
        ref = opipe.process_image(img, keep_planes=False, compat=switches.compat(sw))
        assert board_to_sgf(b) == ref["sgf"], (n, sw)
    # the reference's own scans never make the host run a device pass again (the persistent tail finishes their hysteresis): 18 images
    # in passes of 6; what the phases needed is reported
    hy = det.hysteresis_stats()
    assert hy["passes"] == 3 and hy["redone"] == 0 and min(hy["used_max"]) >= 1, hy
    det.close()


def test_glue_golden_through_c_abi():
    G = load_glue_golden()
    det = Detector(0, 1, 1000, 1000)
    for entry in G["cases"]:
        case, exp = entry["case"], entry["expect"]
        grey = synth_grey(case["w"], case["h"], case["seed"])
        d = det.grid_from_lines(grey, np.array(case["circles"], np.float32).reshape(-1, 3), case["hlines"], case["vlines"],
                                Params(line_threshold=case["threshold"], black_threshold=case["black_thr"],
                                       alignment=case["alignment"]))
        name = case["name"]
        assert (d.found_grid, d.valid_grid, d.board_ready) == (exp["found_grid"], exp["valid_grid"], exp["board_ready"]), name
        np.testing.assert_array_equal(d.hcentres, np.array(exp["hcentres"]), err_msg=name)
        np.testing.assert_array_equal(d.vcentres, np.array(exp["vcentres"]), err_msg=name)
        if exp["valid_grid"]:
            np.testing.assert_array_equal(d.hcentres_complete, np.array(exp["hcentres_complete"]), err_msg=name)
            np.testing.assert_array_equal(d.vcentres_complete, np.array(exp["vcentres_complete"]), err_msg=name)
            assert (d.hspace, d.vspace) == (exp["hspace"], exp["vspace"]), name
        if exp["board_ready"]:
            np.testing.assert_array_equal(d.full_board, np.array(exp["full_board"]), err_msg=name)
            np.testing.assert_array_equal(d.stone_brightnesses, np.array(exp["stone_brightnesses"]), err_msg=name)
            assert d.sgf == exp["sgf"], name
    det.close()


def test_reclassify():
    det = Detector(0, 1, 300, 260)
    img = synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0]
    det.detect_batch([img])
    d2 = det.classify(0, 1, Params(black_threshold=250, alignment=(3, 1)))[0]
    parity.compare_detection(d2, opipe.process_image(img, black_thr=250, alignment=(3, 1)))
    det.close()


def test_device_resident_batch_properties():
    """Benchmark-size, oracle-free properties: device-resident input, 64 diagrams; the recovered 19x19 matrix equals
    the generator's occupancy; results are independent of the pass size and bit-stable run to run."""
    import torch
    imgs, occs = synth.synth_batch(range(1000, 1064))
    t = torch.from_numpy(imgs).cuda()
    det_a, det_b = Detector(0, 64, 1024, 1024), Detector(0, 5, 1024, 1024)
    ba = det_a.detect_device(t)
    bb = det_b.detect_device(t)
    ba2 = det_a.detect_device(t)
    for k in range(64):
        assert ba[k].status == 0
        assert (np.ctypeslib.as_array(ba[k].board) == occs[k]).all()
        assert bytes(ba[k]) == bytes(bb[k]) == bytes(ba2[k])
    det_a.close()
    det_b.close()


def test_headless_cli(tmp_path):
    """python -m img2sgf_amd ex1.jpg out.sgf == the reference's recorded result for ex1."""
    from img2sgf_amd.__main__ import main
    out = tmp_path / "ex1.sgf"
    assert main([os.path.join(GOLDEN, "test_images", "ex1.jpg"), str(out)]) == 0
    assert out.read_text() == EX1_SGF


def test_bench_under_torchrun_single_rank():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, RCCL backend), with one rank:
    process-group init, barrier, max-over-ranks all-reduce and the board all-gather all go through RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(GOLDEN))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--batch", "96", "--pass-size", "32", "--streams", "2", "--roofline-images", "32", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["boards_match_generator"] is True and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["unit"] == "images/s"


def test_capacity_overflow_is_reported_not_truncated():
    """Noise with a Hough-lines threshold of 1: thousands of circles and line peaks.  Whatever the oracle says exceeds a
    capacity of include/i2s.h must come back as I2S_ST_CAPACITY (never silently cut); otherwise the record equals the oracle's."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (1024, 1024), dtype=np.uint8)
    det = Detector(0, 1, 1024, 1024)
    d = det.detect_batch([img], Params(line_threshold=1))[0]
    det.close()
    ref = opipe.process_image(img, threshold=1, keep_planes=True)
    dbg = [cvo.hough_circles(b, debug=True)[1] for b in ref["blurs"]]
    over = (len(ref["circles_all"]) > 16384 or max(len(c) for c in ref["circles_per_variant"]) > 2048
            or max(len(x["est"]) for x in dbg) > 4096 or max(x["n_centers"] for x in dbg) > 1024 * 1024 // 8
            or len(ref["hlines"]) > 1024 or len(ref["vlines"]) > 1024)
    if over:
        assert d.status == 100 and not d.board_ready and d.sgf is None
    else:
        assert d.status != 100
        parity.compare_detection(d, ref)
    # a context one unit too small for a crowded image reports it: see test_capacity_grows_with_the_context_area


def test_capacity_by_lines_first_and_last_in_full_batch():
    """k_line_peaks' overflow left the record's circle count standing while the board record said 0; the packed
    full record was sized from one and written from the other (overrun into the next image's record / past the buffer)."""
    det = Detector(0, 4, 3000, 300)
    parity.check_capacity_by_lines_in_full_batch(det, [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in (0, 1)])
    det.close()
    det = Detector(0, 2, 3000, 300)          # two device passes: the overflowing strip is alone-first in one, last in the other
    parity.check_capacity_by_lines_in_full_batch(det, [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in (2, 3)])
    det.close()


def test_hysteresis_pass_budget_growth():
    """A weak edge that snakes through many tiles and is anchored by a single strong seed needs more hysteresis passes than the
    one plain launch a fresh context starts with: the rest runs inside the persistent tail kernel (grid barriers between passes),
    the result equals the oracle exactly; the second call starts with as many plain launches as the first one needed; and a batch
    of such images mixed with diagrams (uneven load on the tail's workgroups) is still exact."""
    h, w = 256, 512
    img = np.full((h, w), 100, np.uint8)
    # serpentine of low-contrast steps (magnitude 4*30 = 120: weak for 50/200) ...
    for k, y in enumerate(range(20, h - 20, 24)):
        img[y:y + 12, 10:w - 10] = 130
    for k, y in enumerate(range(20, h - 44, 24)):
        x = w - 30 if k % 2 == 0 else 10
        img[y:y + 36, x:x + 20] = 130
    # ... with one strong seed (magnitude > 200) at its start
    img[20:32, 10:14] = 255
    det = Detector(0, 1, w, h)
    parity.run_and_compare(det, [img], internals=False)
    parity.run_and_compare(det, [img], internals=False)
    det.close()
    det = Detector(0, 8, w, h)
    rng = np.random.default_rng(4)
    batch = [img, synth.synth_diagram(1, geom=synth.GEOM_SMALL)[0], img[:, ::-1].copy(), rng.integers(0, 256, (200, 300), dtype=np.uint8),
             img[::-1].copy(), synth.synth_diagram(2, geom=synth.GEOM_SMALL, noisy=True)[0], img.T.copy()[:256, :256], img]
    parity.run_and_compare(det, batch, internals=False)
    det.close()


def test_device_contrast_brightness_all_fixtures():
    """raw decoded RGB in, Pillow's contrast / brightness reproduced on the device (bit-exact), then the full path."""
    from img2sgf_amd import preprocess
    raws = [np.array(preprocess.load_image(os.path.join(GOLDEN, "test_images", n))) for n in IMAGES]
    det = Detector(0, len(raws), max(i.shape[1] for i in raws), max(i.shape[0] for i in raws))
    for (c, b) in [(70, 50), (40, 75)]:
        boards = det.detect_batch(raws, Params(contrast=c, brightness=b), full=False)
        for k, n in enumerate(IMAGES):
            want = preprocess.enhance(preprocess.load_image(os.path.join(GOLDEN, "test_images", n)), c, b)
            np.testing.assert_array_equal(det.fetch_source(k), want, err_msg=n)
            if (c, b) == (70, 50):
                assert board_to_sgf(boards[k]) == opipe.process_image(want, keep_planes=False)["sgf"], n
    det.close()


def test_device_rotate_crop_all_fixtures():
    """SURVEY 8f-1, the rest of crop_and_rotate_image (img2sgf.py:110-114): raw decoded RGB in; Pillow's rotate (NEAREST, white
    fill, the reference's centre) + crop + contrast + brightness all on the device.  The staged source must equal Pillow's
    result bit for bit (Pillow is installed: this row is pinned by the real dependency), the SGF the oracle's."""
    from img2sgf_amd import preprocess
    pils = [preprocess.load_image(os.path.join(GOLDEN, "test_images", n)) for n in IMAGES]
    raws = [np.array(p) for p in pils]
    rng = np.random.default_rng(3)
    det = Detector(0, len(raws), max(i.shape[1] for i in raws) + 16, max(i.shape[0] for i in raws) + 16)
    for round_ in range(3):
        angles = [0.0, 2.0, -1.5][round_] + rng.uniform(-0.5, 0.5, len(raws)) * (round_ > 0)
        sels = []
        for im in raws:
            h, w = im.shape[:2]
            if round_ == 0:
                sels.append(None)
            elif round_ == 1:
                sels.append((int(w * 0.03), int(h * 0.02), w - int(w * 0.02), h - int(h * 0.04)))
            else:
                sels.append((-5, -3, w + 7, h + 2))          # reaches outside: Image.crop pads with 0
        xfs = [preprocess.xform((im.shape[1], im.shape[0]), float(a), s) for im, a, s in zip(raws, angles, sels)]
        boards = det.detect_batch(raws, Params(contrast=70, brightness=50), full=False, xforms=xfs)
        for k, n in enumerate(IMAGES):
            want = preprocess.enhance(pils[k], 70, 50, rotate_angle=float(angles[k]), selection=sels[k])
            np.testing.assert_array_equal(det.fetch_source(k), want, err_msg="%s round %d" % (n, round_))
            ref = opipe.process_image(want, keep_planes=False)
            assert bool(boards[k].status == 0) == bool(ref.get("board_ready")), (n, round_)
            if ref.get("board_ready"):
                assert board_to_sgf(boards[k]) == ref["sgf"], (n, round_)
    det.close()


def test_rotate_crop_device_resident_and_errors():
    import torch
    from PIL import Image
    from img2sgf_amd import preprocess
    img = synth.synth_diagram(9)[0]
    dev = torch.from_numpy(img[None]).cuda()
    det = Detector(0, 1, 1024, 1024)
    xf = preprocess.xform((1024, 1024), 1.0, (12, 8, 1010, 1000))
    boards, _ = det.detect_ptrs([dev.data_ptr()], [1024], [1024], [1024], [1], Params(), True, False, xforms=[xf])
    ref = np.array(Image.fromarray(img).rotate(angle=-1.0, fillcolor="white", center=preprocess.rectangle_centre((12, 8, 1010, 1000)))
                   .crop((12, 8, 1010, 1000)))
    np.testing.assert_array_equal(det.fetch_source(0, 1), ref)
    assert (dev.cpu().numpy()[0] == img).all()                   # the caller's device buffer is never written
    assert board_to_sgf(boards[0]) == opipe.process_image(ref, keep_planes=False)["sgf"]
    with pytest.raises(Exception):                               # region larger than the context
        det.detect_batch([img], Params(), xforms=[preprocess.xform((1024, 1024), 0.0, (-10, 0, 1024, 1024))])
    with pytest.raises(Exception):                               # empty box
        det.detect_batch([img], Params(), xforms=[(preprocess.rotate_matrix(0, (0, 0)), (5, 5, 5, 9))])
    det.close()


def test_scheduled_ragged_batch_all_fixtures():
    """SURVEY 8f-4 (ragged-batch scheduler): 36 real images in shuffled order through 5-image passes formed by area; every
    board record must equal the one the image gets on its own."""
    raws = [opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", n)) for n in IMAGES]
    rng = np.random.default_rng(8)
    order = rng.permutation(2 * len(raws))
    batch = [raws[i % len(raws)] for i in order]
    det = Detector(0, 5, max(i.shape[1] for i in raws), max(i.shape[0] for i in raws))
    single = [bytes(det.detect_batch([r], Params(), full=False)[0]) for r in raws]
    boards = det.detect_batch(batch, Params(schedule=True), full=False)
    for k, i in enumerate(order):
        assert bytes(boards[k]) == single[i % len(raws)], (k, IMAGES[i % len(raws)])
    det.close()


def test_streamed_host_batch_matches_single_stream():
    """StreamedDetector.detect_batch: ragged host images dealt to 3 streams by area-sorted passes; same records, input order."""
    from img2sgf_amd.pipeline import StreamedDetector
    raws = [opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", n)) for n in IMAGES]
    batch = [raws[i % len(raws)] for i in np.random.default_rng(9).permutation(2 * len(raws))]
    mw, mh = max(i.shape[1] for i in raws), max(i.shape[0] for i in raws)
    det = Detector(0, 8, mw, mh)
    want = det.detect_batch(batch, Params(), full=False)
    det.close()
    sd = StreamedDetector(0, 3, 4, mw, mh)
    got = sd.detect_batch(batch, Params())
    sd.close()
    for k in range(len(batch)):
        assert bytes(got[k]) == bytes(want[k]), k


def test_large_images():
    """Sizes beyond the benchmark's: a 2048x2048 mosaic of four diagrams, a 1536x2560 image and a 1100x1700 colour one: every
    plane, circle, line and record against the oracle."""
    a, b2 = synth.synth_diagram(11)[0], synth.synth_diagram(12)[0]
    big = np.ascontiguousarray(np.block([[a, b2], [b2[::-1], a[:, ::-1]]]))
    wide = np.ascontiguousarray(np.pad(big, ((0, 0), (0, 512)), constant_values=255)[:1536])
    sub = wide[:1100, :1700]
    col = np.ascontiguousarray(np.stack([sub, sub[:, ::-1], sub], axis=-1))
    det = Detector(0, 3, 2560, 2048)
    parity.run_and_compare(det, [big, wide, col])
    det.close()


def test_phone_photo_and_extreme_aspect_sizes():
    """Maximum sizes: a 12-megapixel colour image (4032 x 3024, what a phone hands the reference when a user photographs a diagram:
    a mosaic of the reference's largest scan and a synthetic diagram, so coordinates beyond 4095 carry circles, lines and stones)
    and the two extreme aspect ratios a context allows (16384 wide / 16384 tall strips holding a row / column of diagrams' crops):
    coordinates up to 16383 in every packed field (edge records, centre lists, sort keys, erase boxes, rho rows).  Every plane,
    circle, line and record against the oracle."""
    scan = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex16.jpg"))         # 1265 x 1245 RGB
    diag = synth.synth_diagram(21)[0]
    photo = np.full((3024, 4032, 3), 255, np.uint8)
    for k, (y, x) in enumerate([(0, 0), (1500, 2600), (1700, 100)]):
        photo[y:y + scan.shape[0], x:x + scan.shape[1]] = scan if k != 1 else scan[::-1, ::-1]
    for y, x in [(100, 1400), (1990, 1500), (1990, 3000)]:
        photo[y:y + 1024, x:x + 1024] = diag[:, :, None]
    det = Detector(0, 1, 4032, 3024)
    parity.run_and_compare(det, [photo])
    det.close()
    strip = np.full((200, 16384), 255, np.uint8)
    for k in range(16):
        strip[:, k * 1024:(k + 1) * 1024] = synth.synth_diagram(30 + k)[0][300:500]
    strip[:, 16380:] = 0
    det = Detector(0, 1, 16384, 200)
    parity.run_and_compare(det, [strip])
    det.close()
    det = Detector(0, 1, 200, 16384)
    parity.run_and_compare(det, [np.ascontiguousarray(strip.T)])
    det.close()


def test_record_indices_beyond_2_to_31():
    """A context whose edge-record array holds more than 2^31 records (max_batch x 8 inputs x bins x 1024: 72 images of up to
    2048 x 2048 here; 320 images of 1024 x 1024 is the same): k_vote_centres packs a record index and the ray's direction into one
    dword, and until the end of round 4 that index counted from the start of the whole array -- images whose records lay beyond 2^31
    (from image 64 of this context on) silently got wrong accumulators and wrong boards.  The index is now relative to the workgroup's
    pair of inputs.  Small diagrams in the large context against a small context (itself checked against the oracle elsewhere) and
    against the generator."""
    seeds = range(72)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in seeds]
    big = Detector(0, 72, 2048, 2048)
    small = Detector(0, 8, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    a = big.detect_batch(imgs, full=False)
    b = small.detect_batch(imgs, full=False)
    for k in range(len(imgs)):
        assert bytes(a[k]) == bytes(b[k]), k
    for name in ("removed", "edges"):
        np.testing.assert_array_equal(big.fetch_plane(71, name), small.fetch_plane(7, name))
    ref = opipe.process_image(imgs[71])
    assert board_to_sgf(a[71]) == ref["sgf"]
    big.close(); small.close()


def test_largest_batch_and_largest_geometry():
    """The two ends of what i2s_create accepts: 4096 images in ONE device pass (tiny ragged grey / two-valued / colour images) and a
    16384 x 16384 context (262 144 edge bins and 131 072 hysteresis tiles per plane) holding ordinary images -- records byte for byte
    as from small contexts, one image of each against the oracle."""
    rng = np.random.default_rng(5)
    imgs = []
    for k in range(4096):
        h, w = int(rng.integers(1, 64)), int(rng.integers(1, 64))
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 3 == 0 else
                    np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8) if k % 3 == 1 else rng.integers(0, 256, (h, w), dtype=np.uint8))
    big = Detector(0, 4096, 64, 64)
    small = Detector(0, 16, 64, 64)
    a = big.detect_batch(imgs, full=False)
    b = small.detect_batch(imgs, full=False)
    assert [bytes(x) for x in a] == [bytes(x) for x in b]
    big.close()
    parity.run_and_compare(small, imgs[4090:4096])
    small.close()
    huge = Detector(0, 1, 16384, 16384)
    for im in (synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0], opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex9.jpg"))):
        parity.run_and_compare(huge, [im])
    huge.close()


def test_validate_grid_from_none_to_1024_centres():
    det = Detector(0, 1, 600, 600)
    parity.check_validate_grid_capacity(det, 300)
    det.close()


def test_find_lines_on_degenerate_images():
    det = Detector(0, 1, 700, 700)
    parity.check_find_lines_degenerate(det, 700)
    det.close()


def test_capacity_grows_with_the_context_area():
    """4096 small rings on a 16-pixel pitch: two of the ten HoughCircles calls return ~3840 circles each, 7690 in all -- more
    than round 1's fixed lists (2048 per call, 4096 per image) could hold.  The reference's lists are unbounded
    (img2sgf.py:179-186); here the capacities grow with the area the context was created for: the image overflows a 1024 x 1024
    context (one unit per started megapixel; reported as I2S_ST_CAPACITY, never truncated) and completes, equal to the oracle in
    every plane, circle and record, in a 2048 x 2048 context."""
    yy, xx = np.mgrid[0:1024, 0:1024]
    d2 = (yy - ((yy // 16) * 16 + 8)) ** 2 + (xx - ((xx // 16) * 16 + 8)) ** 2
    img = np.where((d2 <= 36) & (d2 >= 16), 0, 255).astype(np.uint8)
    small = Detector(0, 1, 1024, 1024)
    d_small = small.detect_batch([img])[0]
    small.close()
    assert d_small.status == 100 and not d_small.board_ready
    large = Detector(0, 1, 2048, 2048)
    d = parity.run_and_compare(large, [img])[0]
    large.close()
    assert d.status != 100 and len(d.circles_all) > 4096 and max(d.n_per_slot) > 2048


def _blur_planes_match(det, imgs):
    det.detect_batch(imgs, full=False)
    for i, im in enumerate(imgs):
        grey = cvo.bgr2gray(im)
        want = {"grey": grey, "median3": cvo.median_blur(grey, 3), "median5": cvo.median_blur(grey, 5),
                "median7": cvo.median_blur(grey, 7), "gauss3": cvo.gaussian_blur(grey, 3, 3),
                "gauss5": cvo.gaussian_blur(grey, 5, 5), "gauss7": cvo.gaussian_blur(grey, 7, 7)}
        for name, w_ in want.items():
            got = det.fetch_plane(i, name)
            bad = np.argwhere(got != w_)
            assert len(bad) == 0, "%s of image %d %s differs at %d px, first %s got %d want %d" % (
                name, i, im.shape, len(bad), bad[0], got[tuple(bad[0])], w_[tuple(bad[0])])


def test_blur_bank_shapes():
    """The fused register-resident blur kernel (k_blur: 3 Gaussians + 3x3 median, borders as byte permutations of the lane's
    dword triple) and the bit-plane medians on every awkward shape: widths around the 4-pixel lane, the 256-pixel wavefront
    and the 1024-pixel workgroup, heights around the 7-row ring and the 64-row band, grey sources used in place (dword-aligned
    rows) and staged copies (odd widths), colour sources."""
    rng = np.random.default_rng(77)
    shapes = [(1, 1), (1, 2), (2, 1), (3, 3), (1, 9), (9, 1), (4, 4), (5, 7), (6, 8), (7, 13), (8, 8), (9, 12), (63, 255),
              (64, 256), (65, 257), (66, 260), (70, 1023), (71, 1024), (129, 1025), (130, 1028), (5, 1280), (200, 252), (127, 4)]
    det = Detector(0, 4, 1280, 200)
    for k in range(0, len(shapes), 4):
        imgs = []
        for (h, w) in shapes[k:k + 4]:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
            elif kind == 1:
                imgs.append(np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8))      # extremes: 255 * 256 * 256 sums
            else:
                imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        _blur_planes_match(det, imgs)
    det.close()
    # few grey levels: bit planes that repeat their upper neighbour are skipped by the bit-serial medians (med_repeat)
    det = Detector(0, 4, 300, 200)
    for lv in [(0, 255), (0, 128, 255), (0, 64, 192, 255), (17, 17), (0xF0, 0x0F, 0xFF, 0x00), (1, 2, 3), (254, 255)]:
        ims = [np.array(lv, np.uint8)[rng.integers(0, len(lv), (h, w))] for (h, w) in ((200, 300), (77, 130), (9, 9), (150, 57))]
        ims[1] = (ims[1] & 0xF0).astype(np.uint8)
        _blur_planes_match(det, ims)
    det.close()
    # saturated planes (the largest partial sums of the float Gaussians) and device-resident sources used in place
    import torch
    det = Detector(0, 2, 1024, 200)
    for val in (255, 254, 1):
        _blur_planes_match(det, [np.full((100, 1000), val, np.uint8)])
    t = torch.from_numpy(rng.integers(0, 256, (2, 200, 1024), dtype=np.uint8)).cuda()
    det.detect_device(t)
    for i in range(2):
        grey = t[i].cpu().numpy()
        for name, k in (("gauss3", 3), ("gauss5", 5), ("gauss7", 7)):
            np.testing.assert_array_equal(det.fetch_plane(i, name), cvo.gaussian_blur(grey, k, k))
        for name, k in (("median3", 3), ("median5", 5), ("median7", 7)):
            np.testing.assert_array_equal(det.fetch_plane(i, name), cvo.median_blur(grey, k))
        np.testing.assert_array_equal(det.fetch_plane(i, "grey"), grey)
    det.close()


def test_two_valued_band_speculation_restarts():
    """k_blur walks a band as if it held nothing but 0 and 255 and starts over in the general mode at the first other byte:
    0 / 255 images with ONE stray grey pixel -- deep inside a band, in its first and last rows, in the three halo rows it shares
    with the band above / below, in the dword a wavefront borrows from its neighbour, in the last column / row of the image --
    and the value 254 / 1 (an odd bit pattern only in bit 0 / in bits 1 .. 7).  All six planes against the oracle."""
    rng = np.random.default_rng(5)
    h, w = 200, 600                                      # bands of 64 rows (the last one 8 rows), column groups of 256 (the last one 88)
    base = np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8)
    base[40:50, 100:400] = 255; base[120:140, :] = 0
    spots = [(0, 0, 128), (37, 300, 7), (63, 255, 254), (64, 256, 1), (61, 10, 200), (66, 511, 100), (127, 259, 3), (130, 252, 77),
             (191, 599, 254), (199, 599, 128), (199, 0, 1), (192, 512, 129), (100, 255, 126), (2, 257, 64)]
    det = Detector(0, 4, w, h)
    _blur_planes_match(det, [base])                      # no stray pixel: every band stays in the two-valued mode
    for k in range(0, len(spots), 4):
        imgs = []
        for (y, x, v) in spots[k:k + 4]:
            im = base.copy(); im[y, x] = v
            imgs.append(im)
        _blur_planes_match(det, imgs)
    det.close()


def test_small_pass_then_full_pass_on_one_context():
    """Per-context buffers sized on first use (the packed full records' offsets, round 4) are sized for the context's batch, not for
    the pass that comes first: 1 image, then 6, then 2, full records every time, on one context."""
    det = Detector(0, 6, 320, 300)
    imgs = [synth.synth_diagram(s, geom=synth.GEOM_SMALL)[0] for s in range(20, 26)]
    parity.run_and_compare(det, imgs[:1])
    parity.run_and_compare(det, imgs)
    parity.run_and_compare(det, imgs[2:4])
    det.close()


def _canny_maps_match(det, imgs, params):
    """edge image (main Canny) and the eight HoughCircles Canny maps of every image against the oracle's Canny of the same plane."""
    det.detect_batch(imgs, params, full=False)
    hc_lo = max(1, params.hc_param1 // 2)
    for i, im in enumerate(imgs):
        want = cvo.canny(im, params.canny_lo, params.canny_hi)
        got = det.fetch_plane(i, "edges")
        bad = np.argwhere(got != want)
        assert len(bad) == 0, "edges of image %d %s (%s) differ at %d px, first %s" % (i, im.shape, params, len(bad), bad[0])
        for v, name in enumerate(parity.VARIANT_PLANES):
            plane = det.fetch_plane(i, name)
            want = cvo.canny(plane, hc_lo, params.hc_param1)
            got = (det.fetch_plane(i, 9 + 1 + v) == 2).astype(np.uint8) * 255
            bad = np.argwhere(got != want)
            assert len(bad) == 0, "HoughCircles Canny of %s, image %d %s (%s) differs at %d px, first %s" % (
                name, i, im.shape, params, len(bad), bad[0])


def test_canny_byte_walk_shapes_thresholds_and_restarts():
    """Round 4: bands of pure 0 / 255 pixels take Sobel + suppression in bytes (k_sobel_nms_rows, BIN), everything else and every band
    whose walk meets another value in 16-bit pairs -- the second walk overwrites the first.  0 / 255 images on the awkward shapes (the
    4-pixel lane, the 256-pixel wavefront whose end lanes compute a neighbour's magnitude themselves, the 32-row band), thresholds
    around the multiples of 255 the byte walk compares in (and below 0, and above 8 x 255 where nothing passes), fused and separate
    main Canny, and stray grey pixels that stop a walk in its first row, in the middle, in its last rows and in the apron it shares
    with the next band / the next wavefront."""
    rng = np.random.default_rng(404)
    shapes = [(1, 1), (2, 3), (3, 2), (5, 4), (4, 5), (9, 9), (31, 255), (32, 256), (33, 257), (34, 258), (35, 259), (64, 260),
              (70, 511), (40, 513), (66, 1023), (67, 1024), (68, 1025), (100, 1030)]
    imgs = []
    for k, (h, w) in enumerate(shapes):
        dens = (0.5, 0.1, 0.9)[k % 3]
        im = np.where(rng.random((h, w)) < dens, 0, 255).astype(np.uint8)
        if h > 20 and w > 20:                                # some structure: bars, a disc, constant regions
            im[h // 3:h // 3 + 3, :] = 0; im[:, w // 2:w // 2 + 2] = 255
            yy, xx = np.mgrid[0:h, 0:w]
            im[(yy - h // 2) ** 2 + (xx - w // 3) ** 2 < (min(h, w) // 4) ** 2] = 0
        imgs.append(im)
    det = Detector(0, 6, 1032, 100)
    det.set_debug(True)
    param_sets = [Params(), Params(canny_lo=100, canny_hi=200), Params(canny_lo=-5, canny_hi=0, hc_param1=1),
                  Params(canny_lo=254, canny_hi=255, hc_param1=510), Params(canny_lo=255, canny_hi=256, hc_param1=511),
                  Params(canny_lo=509, canny_hi=1020, hc_param1=1021), Params(canny_lo=1275, canny_hi=2039, hc_param1=2040),
                  Params(canny_lo=2040, canny_hi=3000, hc_param1=4100), Params(canny_lo=0, canny_hi=0, hc_param1=3)]
    for pi, prm in enumerate(param_sets):
        for k in range(0, len(imgs), 6):
            if pi < 2 or k % 12 == (pi % 2) * 6:
                _canny_maps_match(det, imgs[k:k + 6], prm)
    # restarts: one stray pixel per image
    h, w = 100, 600
    base = np.where(rng.random((h, w)) < 0.4, 0, 255).astype(np.uint8)
    base[50:60, 100:500] = 0
    spots = [(0, 0, 128), (1, 5, 3), (2, 300, 254), (17, 255, 1), (31, 256, 77), (32, 10, 200), (33, 259, 100), (30, 511, 90),
             (61, 252, 2), (63, 599, 129), (64, 3, 60), (65, 512, 250), (95, 257, 127), (99, 599, 5), (99, 0, 251), (66, 260, 33)]
    for k in range(0, len(spots), 4):
        batch = []
        for (y, x, v) in spots[k:k + 4]:
            im = base.copy(); im[y, x] = v
            batch.append(im)
        _canny_maps_match(det, batch, Params())
        _canny_maps_match(det, batch, Params(canny_lo=100))
    # the same through the colour front end: channels equal (-> the grey plane's walk) and one coloured pixel (-> the 3-plane mode)
    rgb = np.repeat(base[:, :, None], 3, axis=2)
    rgb2 = rgb.copy(); rgb2[40, 300] = (255, 0, 0)
    _canny_maps_match(det, [rgb, rgb2], Params())
    det.close()


def test_plain_rounding_gaussian_taps_use_the_integer_kernels():
    """gauss_kernel_mode = 1 can give tap sums of 257, outside the float kernel's exactness condition: the
    integer kernels take over and still match the oracle."""
    rng = np.random.default_rng(3)
    img = np.where(rng.random((90, 130)) < 0.3, 255, rng.integers(0, 256, (90, 130))).astype(np.uint8)
    det = Detector(0, 1, 130, 90)
    det.detect_batch([img], Params(gauss_kernel_mode=1), full=False)
    for name, k in (("gauss3", 3), ("gauss5", 5), ("gauss7", 7)):
        np.testing.assert_array_equal(det.fetch_plane(0, name), cvo.gaussian_blur(img, k, k, 1))
    np.testing.assert_array_equal(det.fetch_plane(0, "median3"), cvo.median_blur(img, 3))
    det.close()


def test_hysteresis_plain_launch_path_equals_the_persistent_tail(monkeypatch):
    """after ONE grid-barrier timeout of k_hysteresis_tail a context stops using it (i2s_ctx::hy_no_tail: plain launches, the
    tail kernel only reports which pass reached the fixed point).  I2S_HYST_NO_TAIL puts a fresh context on that path: several batches
    of mixed noisy images, serpentines and diagrams must give the same Canny maps, edge images and records as the persistent tail --
    and both must equal the oracle."""
    rng = np.random.default_rng(12)
    h, w = 256, 512
    snake = np.full((h, w), 100, np.uint8)
    for y in range(20, h - 20, 24):
        snake[y:y + 12, 10:w - 10] = 130
    for k, y in enumerate(range(20, h - 44, 24)):
        x = w - 30 if k % 2 == 0 else 10
        snake[y:y + 36, x:x + 20] = 130
    snake[20:32, 10:14] = 255
    batches = []
    for r in range(4):
        batches.append([snake if (r + k) % 3 == 0 else (synth.synth_diagram(r * 8 + k, geom=synth.GEOM_SMALL, noisy=k % 2 == 0)[0] if k % 3 == 1
                        else rng.integers(0, 256, (int(rng.integers(40, 256)), int(rng.integers(40, 512))), dtype=np.uint8)) for k in range(8)])
    tail = Detector(0, 8, w, h)
    monkeypatch.setenv("I2S_HYST_NO_TAIL", "1")
    plain = Detector(0, 8, w, h)
    monkeypatch.delenv("I2S_HYST_NO_TAIL")
    for r, batch in enumerate(batches):
        da = tail.detect_batch(batch, full=False)
        db = plain.detect_batch(batch, full=False)
        for k in range(len(batch)):
            assert bytes(da[k]) == bytes(db[k]), (r, k)
            for plane in ["edges", "removed"] + [9 + m for m in range(9)]:
                np.testing.assert_array_equal(tail.fetch_plane(k, plane), plain.fetch_plane(k, plane), err_msg="batch %d image %d plane %s" % (r, k, plane))
    parity.run_and_compare(plain, batches[0])
    tail.close(); plain.close()


def test_blur_flags_across_passes_of_different_sizes():
    """k_blur's "this band is not two-valued" flags live in a per-context array indexed with the context's band grid and are
    cleared every pass.  One context, passes whose images differ in size: a large 0 / 255 image with stray grey pixels in its LAST band
    row and column (flags at the far edge of the grid), then small images whose bands are all clean, then the large one again, then a
    several-pass batch mixing both -- all six blur planes against the oracle every time."""
    rng = np.random.default_rng(21)
    big = np.where(rng.random((200, 600)) < 0.5, 0, 255).astype(np.uint8)
    big[199, 599] = 128; big[193, 3] = 7; big[5, 597] = 200
    small = np.where(rng.random((70, 90)) < 0.5, 0, 255).astype(np.uint8)
    small_clean_then_grey = small.copy(); small_clean_then_grey[69, 89] = 100
    noisy = rng.integers(0, 256, (130, 300), dtype=np.uint8)
    det = Detector(0, 2, 600, 200)
    for batch in ([big], [small], [big, small], [small, small_clean_then_grey], [noisy], [small], [big, small, noisy, small_clean_then_grey, big]):
        # (_blur_planes_match compares the planes of the LAST device pass: batches longer than max_batch are checked pass by pass)
        for k in range(0, len(batch), 2):
            _blur_planes_match(det, batch[k:k + 2])
    det.close()


def test_c99_host_prints_the_reference_sgf(tmp_path):
    """examples/c_host.c -- a plain C host on the C ABI -- on a PGM of a synthetic diagram and of a reference scan's grey plane: its
    standard output ends with exactly the SGF text the reference's writer produces for the oracle's board (to_SGF, img2sgf.py:781-810)."""
    import subprocess
    from test_abi import _build_c_host
    exe = _build_c_host(tmp_path)
    scan = cvo.bgr2gray(opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex1.jpg")))
    for name, img in (("diagram", synth.synth_diagram(5)[0]), ("ex1", scan)):
        pgm = tmp_path / (name + ".pgm")
        with open(pgm, "wb") as f:
            f.write(b"P5 %d %d 255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
        out = subprocess.run([exe, str(pgm)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        ref = opipe.process_image(img)
        assert ref["board_ready"] and out.stdout.endswith(ref["sgf"]), out.stdout[-400:]
    assert out.stdout.endswith(EX1_SGF)                  # the grey plane of ex1 reads like ex1 itself: the reference's recorded result
