"""The OpenCV-version switches of SURVEY Appendix A.7 as named sets, the same on both sides of a parity check:
`params(name)` is what the product is given (img2sgf_amd.pipeline.Params), `compat(name)` what the oracle is given.

Which OpenCV release each value restates (best knowledge, cv2 is absent here: DESIGN.md section 2a):
  houghlines_numangle  0 = floor(range / theta) + 1 with the "last angle ~ pi" correction   (4.5.2 and later)
                       1 = cvRound(range / theta)                                            (up to 4.5.1; the reference is dated March 2020 = 4.2)
  grey_shift           15 = 3735 / 19235 / 9798                                              (4.x)        14 = 1868 / 9617 / 4899  (3.x)
  gauss_kernel_mode    0 = error-diffused 8-bit taps that sum to 256                         (4.3 / 3.4.10 and later)
                       1 = every tap rounded on its own (sums 257 / 255 / 258)               (4.0 - 4.2)
"""

SWITCH_SETS = {
    "current":          dict(houghlines_numangle=0, grey_shift=15, gauss_kernel_mode=0),
    "numangle_legacy":  dict(houghlines_numangle=1, grey_shift=15, gauss_kernel_mode=0),     # 4.3 .. 4.5.1
    "opencv_4_2":       dict(houghlines_numangle=1, grey_shift=15, gauss_kernel_mode=1),     # the reference's date
    "grey14":           dict(houghlines_numangle=0, grey_shift=14, gauss_kernel_mode=0),
    "gauss_plain":      dict(houghlines_numangle=0, grey_shift=15, gauss_kernel_mode=1),
    "all_alternative":  dict(houghlines_numangle=1, grey_shift=14, gauss_kernel_mode=1),
}
NAMES = list(SWITCH_SETS)


def compat(name):
    """oracle.pipeline.process_image(compat=...)"""
    return dict(SWITCH_SETS[name])


def params_kwargs(name):
    s = SWITCH_SETS[name]
    return dict(houghlines_numangle_mode=s["houghlines_numangle"], grey_shift=s["grey_shift"], gauss_kernel_mode=s["gauss_kernel_mode"])


def params(name, **more):
    from img2sgf_amd.pipeline import Params
    return Params(**dict(params_kwargs(name), **more))
