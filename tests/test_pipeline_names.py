"""The stateless reference-named entry points on the emulated build (CPU) and, marked gpu, on the real library."""
import pytest

import emu_util
import pipeline_names_checks as checks
from img2sgf_amd.pipeline import Detector


@pytest.fixture(scope="module")
def emu_det():
    d = Detector(0, 1, 320, 280, lib=emu_util.emu_library())
    yield d
    d.close()


def test_cluster_lines_emulated(emu_det):
    checks.check_cluster_lines(emu_det)


def test_validate_grid_emulated(emu_det):
    checks.check_validate_grid(emu_det)


def test_find_lines_emulated(emu_det):
    checks.check_find_lines(emu_det)


@pytest.fixture(scope="module")
def gpu_det():
    d = Detector(0, 1, 1024, 1024)
    yield d
    d.close()


@pytest.mark.gpu
def test_cluster_lines_through_c_abi(gpu_det):
    checks.check_cluster_lines(gpu_det)


@pytest.mark.gpu
def test_validate_grid_through_c_abi(gpu_det):
    checks.check_validate_grid(gpu_det)


@pytest.mark.gpu
def test_find_lines_through_c_abi(gpu_det):
    checks.check_find_lines(gpu_det)
