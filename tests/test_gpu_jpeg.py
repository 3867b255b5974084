"""JPEG decode on the device (Huffman stage on the host), pinned against Pillow itself.
The reference enters through Image.open(path).convert("RGB") (img2sgf.py:651); i2s_detect_jpeg_batch must leave exactly those
pixels in the staged source, for the reference's own fixtures (14 sequential, 4 progressive) and for JPEGs Pillow encodes with
every subsampling, quality, Huffman-table, restart-marker and progressive setting; other flavours (CMYK ...) must be refused,
not approximated."""
import io
import os

import numpy as np
import pytest
from PIL import Image

import parity
from helpers import GOLDEN
from img2sgf_amd import preprocess
from img2sgf_amd.pipeline import Detector, I2sError, Params, board_to_sgf
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu

IMAGES = ["ex%d.jpg" % i for i in range(1, 18)] + ["no_circles.jpg"]


def _blob(name):
    with open(os.path.join(GOLDEN, "test_images", name), "rb") as f:
        return f.read()


@pytest.fixture(params=[0, 1, 2], ids=["entropy-host", "entropy-device", "entropy-device-all"])
def on_device(request):
    """i2s_params.jpeg_entropy_device: Huffman decoding on host threads; sequential files in parallel on the device (default);
    those plus one lane per file for the progressive ones."""
    return request.param


def test_all_fixtures_from_file_bytes(on_device):
    """File bytes in, the whole reference flow on the device: decode, contrast 70 / brightness 50, detection."""
    names = list(IMAGES)
    blobs = [_blob(n) for n in names]
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs]
    det = Detector(0, len(names), max(r.shape[1] for r in refs), max(r.shape[0] for r in refs))
    det.detect_jpeg(blobs, Params(jpeg_entropy_device=on_device), full=False)
    for k, n in enumerate(names):
        np.testing.assert_array_equal(det.fetch_source(k, 3), refs[k], err_msg=n)
    boards = det.detect_jpeg(blobs, Params(contrast=70, brightness=50, jpeg_entropy_device=on_device), full=False)
    for k, n in enumerate(names):
        want = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", n))
        np.testing.assert_array_equal(det.fetch_source(k, 3), want, err_msg=n)
        ref = opipe.process_image(want, keep_planes=False)
        assert bool(boards[k].status == 0) == bool(ref.get("board_ready")), n
        if ref.get("board_ready"):
            assert board_to_sgf(boards[k]) == ref["sgf"], n
    det.close()


def test_unsupported_or_broken_files_are_refused(on_device):
    det = Detector(0, 1, 800, 800)
    buf = io.BytesIO()
    Image.fromarray(np.zeros((40, 40, 4), np.uint8), "CMYK").save(buf, "JPEG")
    for blob in (buf.getvalue(), b"\xff\xd8 not a jpeg", _blob("ex9.jpg")[:4000], _blob("ex3.jpg")[:30000]):
        with pytest.raises(I2sError):
            det.detect_jpeg([blob], Params(jpeg_entropy_device=on_device), full=False)
    with pytest.raises(I2sError):
        det.jpeg_info(buf.getvalue())
    assert det.jpeg_info(_blob("ex1.jpg")) == (750, 747, 1)          # progressive, greyscale
    det.close()


def _encode_random(rng):
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
    kind = rng.integers(0, 3)
    if kind == 0:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 3 + yy) % 256, (yy * 2) % 256, (xx + yy * 5) % 256], -1).astype(np.uint8)
    else:
        src = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex%d.jpg" % int(rng.integers(6, 18))))
        img = np.ascontiguousarray(src[:h + 40, :w + 40])
    pil = Image.fromarray(img)
    if rng.random() < 0.3:
        pil = pil.convert("L")
    kw = dict(quality=int(rng.integers(5, 101)), optimize=bool(rng.random() < 0.5), progressive=bool(rng.random() < 0.4))
    if pil.mode == "RGB":
        kw["subsampling"] = int(rng.integers(0, 3))                  # 4:4:4, 4:2:2, 4:2:0
    r = rng.random()
    if r < 0.25:
        kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    elif r < 0.5:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    buf = io.BytesIO()
    try:
        pil.save(buf, "JPEG", **kw)
    except OSError:                       # Pillow's encoder gives up on some option mixes ("Suspension not allowed here")
        buf = io.BytesIO()
        pil.save(buf, "JPEG", quality=kw["quality"])
    blob = buf.getvalue()
    return blob, np.array(Image.open(io.BytesIO(blob)).convert("RGB"))


@pytest.mark.parametrize("seed", range(int(os.environ.get("I2S_JPEG_SEEDS", 24))))
def test_pillow_encoded_images(seed, on_device):
    """Random content and sizes from 1x1, every encoder setting Pillow offers (subsampling, quality, optimised Huffman tables,
    restart intervals, progressive scan scripts); every third seed adds the device rotate / crop on top."""
    rng = np.random.default_rng(7000 + seed)
    pairs = [_encode_random(rng) for _ in range(5)]
    blobs, refs = [p[0] for p in pairs], [p[1] for p in pairs]
    xfs, wants = None, refs
    if seed % 3 == 0:
        xfs, wants = [], []
        for r in refs:
            h, w = r.shape[:2]
            ang = float(rng.uniform(-30, 30))
            x1, y1 = int(rng.integers(-3, 3)), int(rng.integers(-3, 3))
            sel = (x1, y1, max(x1 + 1, w - int(rng.integers(-3, 2))), max(y1 + 1, h - int(rng.integers(-3, 2))))
            xfs.append(preprocess.xform((w, h), ang, sel))
            wants.append(np.array(Image.fromarray(r).rotate(angle=-ang, fillcolor="white", center=preprocess.rectangle_centre(sel)).crop(sel)))
    det = Detector(0, 5, 310, 310)
    dets = det.detect_jpeg(blobs, Params(jpeg_entropy_device=on_device), full=True, xforms=xfs)
    for k, (d, want) in enumerate(zip(dets, wants)):
        np.testing.assert_array_equal(det.fetch_source(k, 3), want, err_msg="image %d" % k)
        if d.status != 100:
            parity.compare_detection(d, opipe.process_image(want))
    det.close()


def test_multi_pass_scheduled_jpeg_batch():
    rng = np.random.default_rng(99)
    blobs = [_encode_random(rng)[0] for _ in range(11)]
    det = Detector(0, 11, 310, 310)
    want = [bytes(b) for b in det.detect_jpeg(blobs, Params(), full=False)]
    det.close()
    det = Detector(0, 4, 310, 310)
    for sched, dev in ((False, 0), (True, 0), (False, 1), (True, 1), (True, 2)):
        got = det.detect_jpeg(blobs, Params(schedule=sched, jpeg_entropy_device=dev), full=False)
        assert [bytes(b) for b in got] == want
    det.close()


def test_headless_cli_mixed_inputs(tmp_path):
    """python -m img2sgf_amd with sequential and progressive JPEGs, all decoded on the device."""
    from img2sgf_amd.__main__ import main
    names = ["ex7.jpg", "ex1.jpg", "ex9.jpg", "ex13.jpg"]          # ex1 is progressive
    assert main([os.path.join(GOLDEN, "test_images", n) for n in names] + ["-o", str(tmp_path)]) == 0
    for n in names:
        want = opipe.process_image(opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", n)), keep_planes=False)
        assert (tmp_path / n.replace(".jpg", ".sgf")).read_text() == want["sgf"], n


def test_streamed_jpeg_batch():
    from img2sgf_amd.pipeline import StreamedDetector
    blobs = [_blob(n) for n in IMAGES] * 2
    det = Detector(0, 8, 1300, 1300)
    want = [bytes(b) for b in det.detect_jpeg(blobs, Params(contrast=70, brightness=50), full=False)]
    det.close()
    sd = StreamedDetector(0, 3, 5, 1300, 1300)
    got = sd.detect_jpeg(blobs, Params(contrast=70, brightness=50))
    sd.close()
    assert [bytes(b) for b in got] == want


def _entropy_span(blob):
    """[first, last) byte of the entropy-coded data of a single-scan file (after the SOS header, before the EOI)."""
    i = blob.index(b"\xff\xda")
    return i + 2 + ((blob[i + 2] << 8) | blob[i + 3]), len(blob) - 2


def test_damaged_entropy_data_same_outcome_on_every_path():
    """Bytes of the entropy-coded data overwritten at random, or a piece cut out: the serial decoder on the host threads is the
    yardstick -- the parallel decoder must refuse exactly the files it refuses and produce the very same pixels for the ones
    it lets through (a damaged stream mostly still decodes, to garbage)."""
    rng = np.random.default_rng(4242)
    det = Detector(0, 1, 420, 420)
    refused = agreed = 0
    for trial in range(120):
        h, w = int(rng.integers(40, 400)), int(rng.integers(40, 400))
        src = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex%d.jpg" % int(rng.integers(6, 18))))
        img = np.ascontiguousarray(src[100:100 + h, 100:100 + w]) if trial % 3 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        kw = dict(quality=int(rng.integers(30, 96)), subsampling=int(rng.integers(0, 3)))
        if trial % 4 == 1:
            kw["restart_marker_rows"] = int(rng.integers(1, 3))
        elif trial % 4 == 2:
            kw["restart_marker_blocks"] = int(rng.integers(2, 40))
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        blob = bytearray(buf.getvalue())
        a, b = _entropy_span(blob)
        if trial % 5 == 4:                                            # a piece missing
            c = int(rng.integers(a, b - 8))
            del blob[c:c + int(rng.integers(1, 64))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                blob[int(rng.integers(a, b))] = int(rng.integers(0, 255))     # never a new 0xFF: no new markers
        blob = bytes(blob)
        out = []
        for mode in (0, 1, 2):
            try:
                det.detect_jpeg([blob], Params(jpeg_entropy_device=mode), full=False)
                out.append(det.fetch_source(0, 3).copy())
            except I2sError:
                out.append(None)
        assert (out[0] is None) == (out[1] is None) == (out[2] is None), "trial %d: refused by some paths only" % trial
        if out[0] is None:
            refused += 1
        else:
            np.testing.assert_array_equal(out[1], out[0], err_msg="trial %d" % trial)
            np.testing.assert_array_equal(out[2], out[0], err_msg="trial %d" % trial)
            agreed += 1
    det.close()
    assert refused > 5 and agreed > 5, (refused, agreed)


def test_large_photograph_many_subsequences():
    """A 2048 x 1536 noisy image: ~1 MB of entropy-coded data, thousands of subsequences per scan; and the same with a restart
    interval of one MCU (tens of thousands of one-subsequence segments)."""
    rng = np.random.default_rng(77)
    yy, xx = np.mgrid[0:1536, 0:2048]
    img = np.stack([(xx * 3 + yy) % 256, (yy * 2 + xx // 3) % 256, (xx + yy * 5) % 256], -1).astype(np.uint8)
    img[400:1200, 300:1500] = rng.integers(0, 256, (800, 1200, 3), dtype=np.uint8)
    det = Detector(0, 1, 2048, 1536)
    for kw in (dict(quality=92, subsampling=2), dict(quality=85, subsampling=0, restart_marker_blocks=1), dict(quality=60, subsampling=1, optimize=True)):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        blob = buf.getvalue()
        want = np.array(Image.open(io.BytesIO(blob)).convert("RGB"))
        det.detect_jpeg([blob], Params(), full=False)
        assert det.jpeg_last_rounds() > 0
        np.testing.assert_array_equal(det.fetch_source(0, 3), want)
    det.close()


def _with_fill_bytes(blob):
    """An extra FF in front of every RSTn marker of the entropy-coded data: legal fill bytes (T.81 B.1.1.2), libjpeg skips them."""
    a, b = _entropy_span(blob)
    out, n = bytearray(blob[:a]), 0
    i = a
    while i < b:
        if blob[i] == 0xFF and 0xD0 <= blob[i + 1] <= 0xD7:
            out += b"\xff" * (1 + n % 3)
            n += 1
        out.append(blob[i])
        i += 1
    return bytes(out + blob[b:]), n


def test_fill_bytes_and_iteration_limit(on_device):
    """Files the parallel decoder hands back: fill bytes in front of restart markers (the serial decoder takes them), and a pass
    that exceeds the iteration's limit (set to one round here)."""
    rng = np.random.default_rng(5150)
    src = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex8.jpg"))
    blobs = []
    for kw in (dict(quality=80, subsampling=2, restart_marker_rows=1), dict(quality=60, subsampling=0, restart_marker_blocks=9), dict(quality=90, subsampling=1)):
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(src[50:350, 80:420])).save(buf, "JPEG", **kw)
        blobs.append(buf.getvalue())
    filled, n = _with_fill_bytes(blobs[0])
    assert n > 5
    filled2, _ = _with_fill_bytes(blobs[1])
    files = [filled, blobs[2], filled2, blobs[0]]
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in files]
    det = Detector(0, len(files), 420, 420)
    det.detect_jpeg(files, Params(jpeg_entropy_device=on_device), full=False)
    for k, r in enumerate(refs):
        np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d" % k)
    assert (det.jpeg_last_rounds() > 0) == (on_device > 0)
    det.jpeg_set_max_rounds(1)
    det.detect_jpeg(files, Params(jpeg_entropy_device=on_device), full=False)
    for k, r in enumerate(refs):
        np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d, one round allowed" % k)
    if on_device:
        # two files had fill bytes (never on the parallel path); the other two were still iterating after the one round
        assert det.jpeg_last_rounds() == 1 and det.jpeg_last_handed_back() == 2
    det.close()


def test_degenerate_streams_go_to_the_serial_decoder_alone(on_device):
    """Round 4: the iteration's limit works per FILE.  A stretch of empty blocks advances one 1024-bit subsequence
    per round (identical blocks never resynchronise a wrong guess).  (a) A page that is blank for its first 800 rows and noise below:
    at the default limit of 48 rounds it alone goes to the serial decoder, the diagrams of the same pass -- done after a handful of
    rounds -- stay on the device; with a limit above its round count nothing is handed back.  (b) A page that is blank altogether
    (under 8 bits per block) never enters the iteration.  Pixels equal Pillow's every time, in every entropy mode."""
    from img2sgf_amd import synth
    rng = np.random.default_rng(808)
    half = np.full((1024, 1024), 255, np.uint8)
    half[800:] = rng.integers(0, 256, (224, 1024), dtype=np.uint8)
    blank = np.full((1024, 1024), 255, np.uint8)

    def encode(imgs):
        out = []
        for im in imgs:
            buf = io.BytesIO()
            Image.fromarray(im).save(buf, "JPEG", quality=85)
            out.append(buf.getvalue())
        return out

    diagrams = [synth.synth_diagram(3, geom=synth.GEOM_SMALL)[0], synth.synth_diagram(4, geom=synth.GEOM_SMALL)[0]]
    det = Detector(0, 3, 1024, 1024)
    for page, handed in ((half, 1), (blank, 0)):
        files = encode([diagrams[0], page, diagrams[1]])
        refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in files]
        det.jpeg_set_max_rounds(48)
        det.detect_jpeg(files, Params(jpeg_entropy_device=on_device), full=False)
        for k, r in enumerate(refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d" % k)
        if on_device:
            assert det.jpeg_last_handed_back() == handed
            assert det.jpeg_last_rounds() == 48 if handed else 0 < det.jpeg_last_rounds() < 20
        det.jpeg_set_max_rounds(2048)
        det.detect_jpeg(files, Params(jpeg_entropy_device=on_device), full=False)
        for k, r in enumerate(refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d, 2048 rounds allowed" % k)
        if on_device:
            assert det.jpeg_last_handed_back() == 0
            assert 48 < det.jpeg_last_rounds() < 400          # (the blank page too: below the limit a short stream is worth trying)
    det.close()


def test_non_interleaved_sequential_scans(on_device):
    """One scan per component in a sequential file (legal, rare; Pillow cannot write it: tests/jpeg_transcode.py rewrites Pillow's
    interleaved scan).  The MCU of such a scan is one block and only blocks that hold samples are coded, so the chroma scans of
    a subsampled image cover fewer blocks than their coefficient arrays have."""
    import jpeg_transcode
    rng = np.random.default_rng(808)
    files = []
    for sub, (h, w) in ((0, (53, 75)), (1, (97, 131)), (2, (120, 203)), (2, (8, 8)), (1, (33, 17))):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx + yy * 2) % 256], -1).astype(np.uint8)
        img[h // 4:h // 2, w // 4:w // 2] = rng.integers(0, 256, (h // 2 - h // 4, w // 2 - w // 4, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=int(rng.integers(40, 95)), subsampling=sub)
        files.append(jpeg_transcode.to_non_interleaved(buf.getvalue(), [2, 0, 1] if sub == 1 else None))
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in files]
    det = Detector(0, len(files), 256, 256)
    det.detect_jpeg(files, Params(jpeg_entropy_device=on_device), full=False)
    for k, r in enumerate(refs):
        np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d" % k)
    det.close()


@pytest.mark.parametrize("schedule", [0, 1])
def test_board_sink_with_more_files_than_one_pass(schedule):
    """i2s.h: with a board sink set, image i's record lands at sink[i] -- also when the JPEG batch needs several device passes
    (each pass calls the ordinary path with pass-local indices) and when the passes are formed by area (schedule)."""
    import ctypes as C
    import torch
    from img2sgf_amd._lib import I2sBoard
    names = ["ex1.jpg", "ex9.jpg", "ex7.jpg", "ex2.jpg", "no_circles.jpg", "ex12.jpg", "ex5.jpg"]
    blobs = [_blob(n) for n in names]
    sizes = [Image.open(io.BytesIO(b)).size for b in blobs]
    det = Detector(0, 3, max(s[0] for s in sizes), max(s[1] for s in sizes))      # 7 files, 3 per pass
    nbytes = C.sizeof(I2sBoard)
    sink = torch.full((len(names) + 1, nbytes), 0xEE, dtype=torch.uint8, device="cuda")
    det.set_board_sink(sink.data_ptr())
    boards = det.detect_jpeg(blobs, Params(schedule=schedule), full=False)
    det.set_board_sink(None)
    torch.cuda.synchronize()
    got = sink.cpu().numpy()
    for k, n in enumerate(names):
        want = np.frombuffer(bytes(boards[k]), dtype=np.uint8)
        np.testing.assert_array_equal(got[k], want, err_msg="sink[%d] (%s)" % (k, n))
    assert (got[len(names)] == 0xEE).all()                                        # nothing past the batch was touched
    det.close()


def _nonconforming_progressive_files():
    """Progressive files whose first passes are NOT disjoint (tests/jpeg_transcode.py::to_progressive): (a) all first passes, clean --
    the control; (b) an AC band coded twice, with different values; (c) the DC scan's band repeated by an AC scan that starts at 0
    is not legal, so: two overlapping luminance bands AND a repeated chroma band; (d) code words whose run carries past the band's
    end into a band a later scan writes.  libjpeg accepts all of them (with warnings) and applies the scans in file order."""
    import jpeg_transcode as T
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:72, 0:88]
    img = np.stack([(xx * 3 + yy * 2) % 256, (xx * yy) % 256, 255 - (xx + yy) % 256], -1).astype(np.uint8)
    img[20:40, 30:60] = rng.integers(0, 256, (20, 30, 3))
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85, subsampling=2)
    base = buf.getvalue()
    clean = [(0, 1, 5, 1), (1, 1, 63, 1), (2, 1, 63, 1), (0, 6, 63, 1)]
    return [T.to_progressive(base, clean),
            T.to_progressive(base, [(0, 1, 5, 1), (0, 3, 9, -1), (1, 1, 63, 1), (2, 1, 63, 1), (0, 6, 63, 1)]),
            T.to_progressive(base, [(0, 1, 9, 1), (1, 1, 63, 1), (0, 6, 63, -1), (2, 1, 63, 1), (1, 1, 20, -1)]),
            T.to_progressive(base, clean, overrun=(0, 3)),
            T.to_progressive(base, [(0, 1, 5, 1), (0, 6, 63, 1), (1, 1, 30, 1), (1, 31, 63, 1), (2, 1, 63, 1)], overrun=(2, -2))]


def test_overlapping_first_passes_decode_in_file_order_on_every_path():
    """the device decodes the leading first passes of a progressive file concurrently, which is only the file's meaning
    while they write disjoint coefficients.  Scans that repeat a band end the device's share (jpeg_parallel), a run that carries past
    its band sends the file to the serial decoder (JPG_REDO): all three entropy paths must give Pillow's pixels, every time."""
    blobs = _nonconforming_progressive_files()
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs]
    assert any((refs[k] != refs[0]).any() for k in (1, 2, 3, 4))           # the defects do change the picture
    det = Detector(0, len(blobs), 96, 80)
    for mode in (0, 1, 2, 1, 1):
        det.detect_jpeg(blobs, Params(jpeg_entropy_device=mode), full=False)
        for k, r in enumerate(refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d, entropy mode %d" % (k, mode))
    # many copies in one pass: more workgroups in flight, more chances for a scheduling-dependent result
    many = [blobs[k % len(blobs)] for k in range(40)]
    det2 = Detector(0, len(many), 96, 80)
    for _ in range(3):
        det2.detect_jpeg(many, Params(jpeg_entropy_device=1), full=False)
        for k in range(len(many)):
            np.testing.assert_array_equal(det2.fetch_source(k, 3), refs[k % len(blobs)], err_msg="copy %d" % k)
    det.close(); det2.close()


def run_handback_beside_redo(make_detector):
    """ONE pass in which sequential files are handed back at the round limit (k_je_pending -> the skip mask of k_je_scan /
    k_je_write, round 5; before, those kernels ran on the files' unconverged states and the host wiped what they wrote) while
    progressive files of the same pass take the JPG_REDO route (a run that carries past its band) and others finish on the device:
    Pillow's pixels for every file at every limit, and the counters say the hand-back happened.  Also driven on the emulated kernels."""
    blobs = _nonconforming_progressive_files()
    src = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex8.jpg"))
    for kw in (dict(quality=80, subsampling=2, restart_marker_rows=1), dict(quality=60, subsampling=0), dict(quality=90, subsampling=1, restart_marker_blocks=7)):
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(src[50:250, 80:330])).save(buf, "JPEG", **kw)
        blobs.append(buf.getvalue())
    blobs = [blobs[k] for k in (5, 3, 0, 6, 4, 1, 7, 2)]                       # interleaved: hand-back candidates between REDO candidates
    refs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs]
    det = make_detector(len(blobs), 256, 208)
    handed = {}
    for limit in (1, 2, 3, 48, 1):
        det.jpeg_set_max_rounds(limit)
        det.detect_jpeg(blobs, Params(jpeg_entropy_device=1), full=False)
        for k, r in enumerate(refs):
            np.testing.assert_array_equal(det.fetch_source(k, 3), r, err_msg="file %d, round limit %d" % (k, limit))
        handed[limit] = det.jpeg_last_handed_back()
        assert det.jpeg_last_rounds() <= limit
    assert handed[1] >= 2 and handed[48] == 0 and handed[1] >= handed[2] >= handed[3], handed
    det.close()


def test_handback_beside_redo_in_one_pass():
    run_handback_beside_redo(lambda nb, w, h: Detector(0, nb, w, h))
