import sys, io, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repo root
import numpy as np
from PIL import Image
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params
from oracle import pipeline as opipe
scan = opipe.load_and_enhance("tests/golden/test_images/ex16.jpg")
diag = synth.synth_diagram(21)[0]
photo = np.full((3024, 4032, 3), 255, np.uint8)
for k, (y, x) in enumerate([(0, 0), (1500, 2600), (1700, 100)]):
    photo[y:y + scan.shape[0], x:x + scan.shape[1]] = scan
for y, x in [(100, 1400), (1990, 1500), (1990, 3000)]:
    photo[y:y + 1024, x:x + 1024] = diag[:, :, None]
rng = np.random.default_rng(1)
photo = np.clip(photo.astype(np.int16) + rng.integers(-20, 20, photo.shape), 0, 255).astype(np.uint8)
cases = {"seq420": dict(quality=92, subsampling=2), "prog444": dict(quality=85, subsampling=0, progressive=True),
         "rst422": dict(quality=95, subsampling=1, restart_marker_rows=1), "grey_prog": dict(quality=90, progressive=True),
         "seq444_opt": dict(quality=98, subsampling=0, optimize=True)}
det = Detector(0, 2, 4032, 3024)
for name, kw in cases.items():
    pil = Image.fromarray(photo)
    if name.startswith("grey"): pil = pil.convert("L")
    buf = io.BytesIO(); pil.save(buf, "JPEG", **kw); blob = buf.getvalue()
    want = np.array(Image.open(io.BytesIO(blob)).convert("RGB"))
    for mode in (0, 1, 2) if name in ("seq420", "grey_prog") else (1,):
        if mode == 2 and len(blob) > 3_000_000: continue
        b = det.detect_jpeg([blob, blob], Params(jpeg_entropy_device=mode), full=False)
        got = det.fetch_source(1, 3)
        print(name, "mode", mode, "bytes", len(blob), "equal", bool((got == want).all()), "rounds", det.jpeg_last_rounds(), "handed", det.jpeg_last_handed_back(), "status", b[0].status, flush=True)
det.close()
