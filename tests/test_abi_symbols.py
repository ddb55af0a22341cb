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


def test_binding_constants_equal_the_header():
    """Every numeric #define / enum value of the header that the Python binding repeats has the same value there."""
    text = open(os.path.join(REPO, "include", "slam_amd.h")).read()
    values = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(SLAM_[A-Z0-9_]+)\s+(-?(?:0x[0-9A-Fa-f]+|\d+))\b", text)}
    values.update({m.group(1): int(m.group(2)) for m in re.finditer(r"\b(SLAM_E[A-Z]+)\s*=\s*(-?\d+)", text)})
    pairs = {"SLAM_SAMPLER_CPYTHON": "SAMPLER_CPYTHON", "SLAM_SAMPLER_COUNTER": "SAMPLER_COUNTER", "SLAM_REC_COUNT": "REC_COUNT",
             "SLAM_REC_ERRSUM": "REC_ERRSUM", "SLAM_REC_INDEX": "REC_INDEX", "SLAM_REC_F": "REC_F", "SLAM_REC_LEN": "REC_LEN",
             "SLAM_SCORE_ALGEBRAIC": "SCORE_ALGEBRAIC", "SLAM_SCORE_SAMPSON": "SCORE_SAMPSON", "SLAM_K_SOLVE": "K_SOLVE",
             "SLAM_K_SCORE": "K_SCORE", "SLAM_K_SELECT": "K_SELECT", "SLAM_K_TRIANGULATE": "K_TRIANGULATE"}
    checked = 0
    for c_name, py_name in pairs.items():
        if c_name in values and hasattr(_lib, py_name):
            assert values[c_name] == getattr(_lib, py_name), c_name
            checked += 1
    assert checked >= 10


def test_library_was_built_from_this_tree():
    """slam_build_hash() of the loaded library equals the hash of the sources in the tree: a stale .so fails here."""
    assert _lib.source_hash() is not None
    assert _lib.build_hash() == _lib.source_hash()
