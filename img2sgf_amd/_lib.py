"""ctypes binding of the C ABI in include/i2s.h (libi2s_hip.so).

The product path has NO CPU fallback: `load()` raises if the HIP library is missing or cannot be
loaded, and `i2s_create` fails when no GPU is visible.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libi2s_hip.so")

BOARD_SIZE = 19
NSLOTS = 10
MAX_CIRCLES = 16384
MAX_LINES = 1024
MAX_CENTRES = 1024
PLANE_NAMES = {"grey": 0, "edges": 1, "median3": 2, "gauss3": 3, "median5": 4, "gauss5": 5, "median7": 6,
               "gauss7": 7, "removed": 8, "canny_map": 9}

STATUS_TEXT = {
    0: "board ready", 1: "no horizontal grid lines found", 2: "only one horizontal grid line",
    3: "horizontal grid lines too close together", 4: "horizontal grid too wide (extra lines?)",
    5: "no vertical grid lines found", 6: "only one vertical grid line",
    7: "vertical grid lines too close together", 8: "vertical grid too wide (extra lines?)",
    9: "too many vertical lines", 10: "too many horizontal lines", 100: "capacity exceeded",
}


class I2sParams(C.Structure):
    _fields_ = [
        ("canny_lo", C.c_int32), ("canny_hi", C.c_int32),
        ("hc_min_dist", C.c_float),
        ("hc_param1", C.c_int32), ("hc_param2", C.c_int32),
        ("hc_min_radius", C.c_int32), ("hc_max_radius", C.c_int32),
        ("line_threshold", C.c_int32), ("black_threshold", C.c_int32),
        ("align_x", C.c_int32), ("align_y", C.c_int32),
        ("min_grid_spacing", C.c_double), ("big_space_ratio", C.c_double), ("angle_tolerance_deg", C.c_double),
        ("grey_shift", C.c_int32), ("gauss_kernel_mode", C.c_int32), ("houghlines_numangle_mode", C.c_int32),
        ("inputs_on_device", C.c_int32),
        ("contrast", C.c_int32), ("brightness", C.c_int32),
        ("schedule", C.c_int32), ("jpeg_entropy_device", C.c_int32),
    ]


class I2sBoard(C.Structure):
    _fields_ = [
        ("board", (C.c_uint8 * BOARD_SIZE) * BOARD_SIZE),
        ("status", C.c_uint8), ("side_to_move", C.c_uint8), ("hsize", C.c_uint8), ("vsize", C.c_uint8),
        ("found_grid", C.c_uint8), ("valid_grid", C.c_uint8), ("pad0", C.c_uint8),
        ("n_black", C.c_uint16), ("n_white", C.c_uint16), ("n_circles", C.c_uint16), ("line_threshold", C.c_uint16),
        ("pad", C.c_uint8 * 8),
    ]


class I2sResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("line_threshold", C.c_int32),
        ("found_grid", C.c_int32), ("valid_grid", C.c_int32), ("board_ready", C.c_int32),
        ("hsize", C.c_int32), ("vsize", C.c_int32),
        ("n_circles", C.c_int32), ("n_per_slot", C.c_int32 * NSLOTS), ("n_circles_kept", C.c_int32),
        ("n_hlines", C.c_int32), ("n_vlines", C.c_int32),
        ("n_hcentres", C.c_int32), ("n_vcentres", C.c_int32),
        ("n_hcomplete", C.c_int32), ("n_vcomplete", C.c_int32),
        ("n_stones", C.c_int32), ("n_black", C.c_int32), ("n_white", C.c_int32), ("side_to_move", C.c_int32),
        ("pad0", C.c_int32),
        ("hspace", C.c_double), ("vspace", C.c_double),
        ("hcentres", C.c_double * MAX_CENTRES), ("vcentres", C.c_double * MAX_CENTRES),
        ("hcentres_complete", C.c_double * MAX_CENTRES), ("vcentres_complete", C.c_double * MAX_CENTRES),
        ("brightness", C.c_double * (BOARD_SIZE * BOARD_SIZE)),
        ("hlines", C.c_float * MAX_LINES), ("vlines", C.c_float * MAX_LINES),
        ("circles", (C.c_float * 3) * MAX_CIRCLES),
        ("circle_kept", C.c_uint8 * MAX_CIRCLES),
        ("detected", (C.c_uint8 * BOARD_SIZE) * BOARD_SIZE),
        ("board", (C.c_uint8 * BOARD_SIZE) * BOARD_SIZE),
        ("pad1", C.c_uint8 * 2),
    ]


class I2sXform(C.Structure):
    """i2s_xform: Pillow's inverse affine matrix of Image.rotate() + the crop box (crop_and_rotate_image, img2sgf.py:110-114)."""
    _fields_ = [("affine", C.c_double * 6), ("crop", C.c_int32 * 4)]


assert C.sizeof(I2sBoard) == 384
assert C.sizeof(I2sResult) == 73384 + (16384 - 4096) * 13 + 4 * (1024 - 256) * 8

EXPORTS = ["i2s_abi_version", "i2s_device_arch", "i2s_default_params", "i2s_choose_threshold", "i2s_strerror", "i2s_last_error",
           "i2s_create", "i2s_destroy", "i2s_detect_batch", "i2s_detect_batch_xf", "i2s_jpeg_info", "i2s_detect_jpeg_batch", "i2s_jpeg_last_rounds", "i2s_jpeg_last_handed_back", "i2s_jpeg_set_max_rounds", "i2s_jpeg_last_timing",
           "i2s_classify_batch", "i2s_grid_from_lines", "i2s_validate_grid", "i2s_find_lines",
           "i2s_fetch_plane", "i2s_fetch_source", "i2s_last_timing", "i2s_set_debug", "i2s_fetch_circle_acc", "i2s_fetch_line_acc",
           "i2s_comm_unique_id", "i2s_comm_create", "i2s_comm_destroy", "i2s_comm_last_error", "i2s_comm_shard", "i2s_comm_all",
           "i2s_set_board_sink", "i2s_allgather_boards",
           "i2s_set_profiling", "i2s_last_kernel_timing", "i2s_kernel_timing_name", "i2s_blur_band_stats", "i2s_hysteresis_stats"]
NSEG = 14
ABI_VERSION = 4
COMM_ID_BYTES = 128


class I2sError(RuntimeError):
    pass


class I2sLibrary:
    """A loaded C-ABI library with typed entry points."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise I2sError(
                "HIP library %s not found: build it with `python -m img2sgf_amd.build` "
                "(there is no CPU fallback)" % path)
        self.path = path
        L = self.dll = C.CDLL(path)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise I2sError("%s does not export %s" % (path, name))
        vp, ip, u8p, f32p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
        L.i2s_abi_version.restype = C.c_int
        L.i2s_default_params.argtypes = [C.POINTER(I2sParams)]
        L.i2s_default_params.restype = None
        L.i2s_choose_threshold.argtypes = [C.c_int, C.c_int]
        L.i2s_strerror.argtypes = [C.c_int]
        L.i2s_strerror.restype = C.c_char_p
        L.i2s_last_error.argtypes = [vp]
        L.i2s_last_error.restype = C.c_char_p
        L.i2s_device_arch.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.i2s_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
        L.i2s_destroy.argtypes = [vp]
        L.i2s_destroy.restype = None
        L.i2s_detect_batch.argtypes = [vp, C.c_int, C.POINTER(vp), ip, ip, ip, ip, C.POINTER(I2sParams),
                                       C.POINTER(I2sBoard), C.POINTER(I2sResult)]
        L.i2s_detect_batch_xf.argtypes = [vp, C.c_int, C.POINTER(vp), ip, ip, ip, ip, C.POINTER(I2sXform),
                                          C.POINTER(I2sParams), C.POINTER(I2sBoard), C.POINTER(I2sResult)]
        L.i2s_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, ip, ip, ip]
        L.i2s_jpeg_last_rounds.argtypes = [vp]
        L.i2s_jpeg_last_handed_back.argtypes = [vp]
        L.i2s_jpeg_set_max_rounds.argtypes = [vp, C.c_int]
        L.i2s_jpeg_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
        L.i2s_detect_jpeg_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(I2sXform),
                                            C.POINTER(I2sParams), C.POINTER(I2sBoard), C.POINTER(I2sResult)]
        L.i2s_classify_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(I2sParams), C.POINTER(I2sBoard),
                                         C.POINTER(I2sResult)]
        L.i2s_grid_from_lines.argtypes = [vp, u8p, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int,
                                          C.POINTER(I2sParams), C.POINTER(I2sBoard), C.POINTER(I2sResult)]
        L.i2s_validate_grid.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, f32p, C.c_int,
                                        C.POINTER(I2sParams), C.POINTER(I2sResult)]
        L.i2s_find_lines.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_size_t, C.POINTER(I2sParams), f32p, ip, f32p, ip]
        L.i2s_fetch_plane.argtypes = [vp, C.c_int, C.c_int, u8p, C.c_size_t]
        L.i2s_fetch_source.argtypes = [vp, C.c_int, u8p, C.c_size_t]
        L.i2s_last_timing.argtypes = [vp, f32p]
        L.i2s_set_debug.argtypes = [vp, C.c_int]
        L.i2s_fetch_circle_acc.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.i2s_fetch_line_acc.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_size_t, ip, ip]
        L.i2s_comm_unique_id.argtypes = [u8p]
        L.i2s_comm_create.argtypes = [C.POINTER(vp), C.c_int, u8p, C.c_int, C.c_int, C.c_int]
        L.i2s_comm_destroy.argtypes = [vp]
        L.i2s_comm_destroy.restype = None
        L.i2s_comm_last_error.argtypes = [vp]
        L.i2s_comm_last_error.restype = C.c_char_p
        L.i2s_comm_shard.argtypes = [vp]
        L.i2s_comm_shard.restype = vp
        L.i2s_comm_all.argtypes = [vp]
        L.i2s_comm_all.restype = vp
        L.i2s_set_board_sink.argtypes = [vp, vp]
        L.i2s_allgather_boards.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        L.i2s_set_profiling.argtypes = [vp, C.c_int]
        L.i2s_blur_band_stats.argtypes = [vp, ip, ip]
        L.i2s_hysteresis_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), ip]
        L.i2s_last_kernel_timing.argtypes = [vp, f32p]
        L.i2s_kernel_timing_name.argtypes = [C.c_int]
        L.i2s_kernel_timing_name.restype = C.c_char_p
        if L.i2s_abi_version() != ABI_VERSION:
            raise I2sError("ABI version mismatch in %s" % path)


def device_arch(lib, ctx):
    """i2s_device_arch: "gfx950:sramecc+:xnack-" on an MI355X; the CPU emulation of tests/emu answers "emulated"."""
    buf = C.create_string_buffer(256)
    rc = lib.dll.i2s_device_arch(ctx, buf, 256)
    if rc != 0:
        raise I2sError("i2s_device_arch: %s" % lib.dll.i2s_strerror(rc).decode())
    return buf.value.decode()


_DEFAULT = None


def load():
    """The product library (libi2s_hip.so next to this file).  Raises I2sError if it is missing.
    I2S_LIBRARY in the environment names another BUILD of the same library (a file called libi2s_hip.so with the same ABI version, e.g.
    an experiment of tools/experiments/ under build/exp/): the GPU suite and the tools then run on that build.  It is never a fallback:
    the named file must exist, load and export every symbol of include/i2s.h like the product does."""
    global _DEFAULT
    if _DEFAULT is None:
        path = os.environ.get("I2S_LIBRARY") or LIB_PATH
        if os.path.basename(path) != os.path.basename(LIB_PATH):
            raise I2sError("I2S_LIBRARY must name a build of %s, got %s" % (os.path.basename(LIB_PATH), path))
        if os.path.realpath(path) != os.path.realpath(LIB_PATH):
            # never silent: a variable left over from tools/experiments/ab.sh or tools/gpu_suite_on_emulator.sh redirects every
            # consumer of the package (CLI, gui_adapter, bench), so each process says once which file it runs on
            import sys
            sys.stderr.write("img2sgf_amd: I2S_LIBRARY is set -- running on %s, NOT the product library %s\n" % (path, LIB_PATH))
        _DEFAULT = I2sLibrary(path)
    return _DEFAULT
