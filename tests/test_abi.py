"""The C-ABI library loads (no GPU needed) and exports every symbol include/maskplanner_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "maskplanner_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("mp_fps_f32", "mp_ball_query_f32", "mp_group_f32", "mp_knn_f32", "mp_knn_bwd_f32",
                 "mp_padded_lengths_f32", "mp_mask_match_f32", "mp_index_points_f32"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from maskplanner_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in maskplanner_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    assert lib.mp_abi_version() == _lib.ABI_VERSION
    assert lib.mp_error_string(-2).decode().startswith("size outside")


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (safe on a GPU-less box)."""
    from maskplanner_amd import _lib
    lib = _lib.load()
    assert lib.mp_fps_f32(None, 2, 100, 10, None, None, None, None) == _lib.MP_EINVAL
    assert lib.mp_fps_f32(None, 0, 100, 10, None, None, None, None) == _lib.MP_OK  # empty batch
    assert lib.mp_ball_query_f32(None, None, 1, 10, 1, 0.2, 0, None, None) == _lib.MP_EINVAL
    assert lib.mp_knn_f32(None, None, None, None, 1, 5, 5, 3, 9, None, None, None, 0, None) == _lib.MP_EINVAL


def test_ops_refuse_cpu_tensors():
    import torch
    from maskplanner_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fps(torch.zeros(1, 10, 3), 2, torch.zeros(1, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.knn(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
