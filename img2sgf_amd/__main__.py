"""Headless counterpart of the reference's command line (img2sgf.py:1256-1269):

    python -m img2sgf_amd input.jpg [output.sgf]

opens the image, applies the reference's default settings (img2sgf.py:616-640: contrast 70, brightness 50, no rotation,
full-image selection, Hough threshold from choose_threshold, black threshold 128, alignment left/top), runs the board
detection on the GPU and writes what the reference's "save" button writes (to_SGF, :781-822).  Several inputs may be
given with -o DIR; they are detected as one batch."""
import argparse
import os
import sys

from . import pipeline, preprocess


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m img2sgf_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("inputs", nargs="+", help="image file(s); with one input a second positional argument is the output .sgf")
    ap.add_argument("-o", "--outdir", help="directory for <name>.sgf when several inputs are given")
    ap.add_argument("--contrast", type=int, default=preprocess.CONTRAST_DEFAULT)
    ap.add_argument("--brightness", type=int, default=preprocess.BRIGHTNESS_DEFAULT)
    ap.add_argument("--rotate", type=float, default=0.0, help="rotation angle in degrees (the reference's rotate slider)")
    ap.add_argument("--selection", type=int, nargs=4, metavar=("X1", "Y1", "X2", "Y2"), help="region to process (default: whole image)")
    ap.add_argument("--threshold", type=int, default=0, help="Hough-lines threshold (0 = choose_threshold)")
    ap.add_argument("--black-threshold", type=int, default=128)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--opencv", metavar="VERSION|auto", help="restate this OpenCV release's arithmetic (e.g. 4.2.0, 4.8.1), or 'auto': "
                    "probe the cv2 installed next to this package for what it computes (Params.from_cv2); default: the package "
                    "defaults = OpenCV 4.3 .. 4.5.1 (Params.opencv_switches).  The active switch set is printed on stderr.")
    args = ap.parse_args(argv)
    inputs, out_single = args.inputs, None
    if len(inputs) == 2 and inputs[1].lower().endswith(".sgf") and not args.outdir:
        inputs, out_single = inputs[:1], inputs[1]
    import numpy as np
    if args.opencv == "auto":
        try:
            import cv2
        except ImportError:
            ap.error("--opencv auto needs an importable cv2 to probe")
        probed = True
        try:
            switches = pipeline.probe_cv2_switches(cv2)
        except Exception as e:      # a build whose Gaussian / HoughLines takes a path the probe does not know (HAL, IPP), or a cv2.error
            probed = False
            print("--opencv auto: the behavioural probe failed on cv2 %s (%s: %s); falling back to the release table for that version"
                  % (cv2.__version__, type(e).__name__, e), file=sys.stderr)
            switches = pipeline.Params.opencv_switches(cv2.__version__)
    else:
        switches = pipeline.Params.opencv_switches(args.opencv) if args.opencv else {}
    params = pipeline.Params(line_threshold=args.threshold, black_threshold=args.black_threshold,
                             contrast=args.contrast, brightness=args.brightness, schedule=True, **switches)
    print("OpenCV switch set: %s (%s)" % (params.switch_set(), ("probed from cv2 " if probed else "release table for cv2 ") + cv2.__version__ if args.opencv == "auto" else
                                          "release " + args.opencv if args.opencv else "package defaults; --opencv VERSION|auto to change"),
          file=sys.stderr)
    # Huffman-coded JPEGs (sequential or progressive) are decoded on the GPU straight from the file bytes (bit-exact with
    # Pillow's decoder); anything else (PNG, CMYK JPEG, ...) is opened with Pillow as the reference does (img2sgf.py:651).  Rotate / crop / contrast /
    # brightness run on the GPU either way.
    blobs, sizes = {}, []
    for k, path in enumerate(inputs):
        with open(path, "rb") as f:
            data = f.read()
        try:
            w, h, _ = pipeline.jpeg_info(data)
            blobs[k] = data
        except pipeline.I2sError:
            im = np.array(preprocess.load_image(path))
            blobs[k] = im
            w, h = im.shape[1], im.shape[0]
        sizes.append((w, h))
    xforms = None
    if args.rotate != 0 or args.selection:
        xforms = [preprocess.xform(sz, args.rotate, args.selection) for sz in sizes]
    out_w = [x[1][2] - x[1][0] for x in xforms] if xforms else [sz[0] for sz in sizes]
    out_h = [x[1][3] - x[1][1] for x in xforms] if xforms else [sz[1] for sz in sizes]
    det = pipeline.Detector(args.device, min(len(inputs), 16), max(out_w), max(out_h))
    results = [None] * len(inputs)
    for is_jpeg in (True, False):
        idx = [k for k in range(len(inputs)) if isinstance(blobs[k], bytes) == is_jpeg]
        if not idx:
            continue
        xf = [xforms[k] for k in idx] if xforms else None
        items = [blobs[k] for k in idx]
        dets = det.detect_jpeg(items, params, xforms=xf) if is_jpeg else det.detect_batch(items, params, xforms=xf)
        for k, d in zip(idx, dets):
            results[k] = d
    rc = 0
    for path, d in zip(inputs, results):
        name = os.path.splitext(os.path.basename(path))[0] + ".sgf"
        out = out_single or (os.path.join(args.outdir, name) if args.outdir else None)
        if not d.board_ready:
            print("%s: board not detected (%s)" % (path, d.status_text), file=sys.stderr)
            rc = 1
            continue
        print("%s: %dx%d board, %d black + %d white stones, %s to play" % (
            path, d.hsize, d.vsize, d.num_black_stones, d.num_white_stones, "black" if d.side_to_move == 1 else "white"),
            file=sys.stderr)
        if out:
            with open(out, "w") as f:
                f.write(d.sgf)
        else:
            sys.stdout.write(d.sgf)
    det.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
