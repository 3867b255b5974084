"""Pre-processing as the reference does it with Pillow (img2sgf.py:651, 110-114, 136-150): open + convert to RGB,
rotate/crop (identity at the defaults), contrast and brightness enhancement, numpy array = the `input_image_np` the
detection starts from.

`enhance()` is the host (Pillow) form.  `xform()` prepares the same rotate + crop for the GPU: it builds the inverse affine
matrix exactly as Pillow's Image.rotate() does in Python and hands it, with the crop box, to i2s_detect_batch_xf, whose
kernel restates Pillow's fixed-point nearest-neighbour transform (csrc/k_preprocess.h); contrast / brightness then run on
the device as well (Params.contrast / .brightness)."""
import math

import numpy as np

CONTRAST_DEFAULT = 70      # img2sgf.py:56
BRIGHTNESS_DEFAULT = 50    # img2sgf.py:57


def load_image(path):
    """img2sgf.py:651."""
    from PIL import Image
    return Image.open(path).convert("RGB")


def enhance(image_pil, contrast=CONTRAST_DEFAULT, brightness=BRIGHTNESS_DEFAULT, rotate_angle=0, selection=None):
    """img2sgf.py:110-114 (crop_and_rotate_image) and 136-150.  selection = (x1, y1, x2, y2) or None for the full image."""
    from PIL import ImageEnhance
    w, h = image_pil.size
    sel = tuple(selection) if selection is not None else (0, 0, w, h)
    centre = ((sel[0] + sel[2]) / 2, sel[1] + sel[3] / 2)          # rectangle_centre, reproduced as written (:106-107)
    region = image_pil.rotate(angle=-rotate_angle, fillcolor="white", center=centre).crop(sel)
    region = ImageEnhance.Contrast(region).enhance(102 / (101 - contrast) - 1)
    region = ImageEnhance.Brightness(region).enhance(450 / (200 - brightness) - 2)
    return np.array(region)


def rectangle_centre(a):
    """img2sgf.py:106-107, reproduced as written (the y term is a[1] + a[3] / 2)."""
    return ((a[0] + a[2]) / 2, a[1] + a[3] / 2)


def rotate_matrix(angle, centre):
    """The inverse affine matrix PIL.Image.Image.rotate(angle, center=centre) passes to transform(AFFINE) (Image.py: the
    expand=False, translate=None branch; with a centre given Pillow takes no 0/90/180/270 shortcut)."""
    angle = angle % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = centre
    x, y = -cx, -cy
    m[2], m[5] = m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]
    m[2] += cx
    m[5] += cy
    return m


def xform(size, rotate_angle=0, selection=None):
    """(affine, crop) for Detector.detect_batch(..., xforms=[...]): crop_and_rotate_image() (img2sgf.py:110-114) for an
    image of `size` = (w, h).  selection = (x1, y1, x2, y2) or None for the full image."""
    w, h = size
    sel = tuple(selection) if selection is not None else (0, 0, w, h)
    crop = tuple(int(round(v)) for v in sel)                       # Image.crop() rounds its box the same way
    return rotate_matrix(-rotate_angle, rectangle_centre(sel)), crop


def load_and_enhance(path, contrast=CONTRAST_DEFAULT, brightness=BRIGHTNESS_DEFAULT):
    return enhance(load_image(path), contrast, brightness)
