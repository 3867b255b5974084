"""Host-side pre-processing exactly as the reference does it with Pillow (img2sgf.py:651, 110-114, 136-150):
open + convert to RGB, rotate/crop (identity at the defaults), contrast and brightness enhancement, numpy array.
The result is the `input_image_np` the GPU pipeline starts from."""
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


def load_and_enhance(path, contrast=CONTRAST_DEFAULT, brightness=BRIGHTNESS_DEFAULT):
    return enhance(load_image(path), contrast, brightness)
