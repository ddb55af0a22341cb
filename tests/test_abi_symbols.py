"""The C-ABI library loads without a GPU and exports every entry point include/slam_amd.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from slam_amd import _lib


def declared_functions():
    text = open(os.path.join(REPO, "include", "slam_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("slam_create", "slam_eightpoint", "slam_ransac_f", "slam_ransac_f_dev", "slam_triangulate",
                 "slam_pose_from_E", "slam_init_pair", "slam_sample_indices", "slam_merge_records"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_library_contains_gfx950_code_only():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in blob


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.slam_num_hypotheses(8, 1000) == 1
    assert lib.slam_num_hypotheses(9, 1000) == 9
    assert lib.slam_num_hypotheses(12, 1000) == 495
    assert lib.slam_num_hypotheses(13, 1000) == 1000
    assert lib.slam_num_hypotheses(4096, 2048) == 2048
    assert lib.slam_num_hypotheses(1 << 30, 1 << 40) == 1 << 40
    assert lib.slam_num_hypotheses(7, 1000) == 0


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SlamError):
        _lib.Context(0)
