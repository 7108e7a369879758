"""CPU: libojf.so loads without a GPU and exports exactly the symbols include/ojf.h declares;
calls that need a device fail loudly instead of falling back to anything."""
import os
import re

import pytest

from online_joint_depthfusion_and_semantic_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'ojf.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ojf_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail('libojf.so is not built: run `python -c "import __graft_entry__ as g; g.build()"`')
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == declared  # the ctypes table and the header agree


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.ojf_version().decode().startswith('ojf ')
    assert isinstance(lib.ojf_last_error(), bytes)
    assert lib.ojf_net_layer_count(3, 9, 5, 0) == 57
    assert lib.ojf_net_layer_count(3, 9, 5, 1) == 85
    assert lib.ojf_net_layer_count(2, 9, 5, 1) == 57
    assert lib.ojf_net_layer_count(1, 9, 5, 0) < 0


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible: the loud-failure path is for GPU-less hosts')
    from online_joint_depthfusion_and_semantic_amd import ops
    with pytest.raises(_lib.OjfError):
        _lib.require_gpu()
    with pytest.raises(_lib.OjfError):
        ops.IntegrateWorkspace((8, 8, 8), 4, 4, 7, ops.MODE_FAST, 'cpu')


def test_argument_validation_without_device():
    lib = _lib.load()
    # null pointers are rejected before any HIP call
    assert lib.ojf_extract(None, None, None, None, 0.02, None, None, 8, 8, 8, 4, 4, 9, -0.1, None, None, 9, 0,
                           None, None, None, None, None) != 0
    assert b'null' in lib.ojf_last_error()
    assert lib.ojf_integrate_workspace_bytes(0, 8, 8, 4, 4, 7, 0) == 0
    # header + 4 B/voxel head table + one 32-B record per entry + touched list
    # FAST: header + head table + (32-B records + 4-B first touches) x tiles x (2048 + 128 * n_tail * 8) + tile counts (16 x 8 pixel tiles)
    assert lib.ojf_integrate_workspace_bytes(8, 8, 8, 4, 4, 7, 0) == 512 + 512 * 4 + 1 * (2048 + 7168) * 36 + 1 * 4


def test_segconv_and_mesh_argument_validation_without_device():
    """The SEGCONV / mesh entry points reject bad descriptions before they touch the device."""
    import ctypes
    import numpy as np
    lib = _lib.load()
    h = ctypes.c_void_p()
    w = np.ones((4, 8, 3, 3), dtype=np.float32)
    assert lib.ojf_segconv_create(ctypes.byref(h), None, None, None, 8, 4, 3, 1, 1, 1) != 0 and b'null' in lib.ojf_last_error()
    assert lib.ojf_segconv_create(ctypes.byref(h), w.ctypes.data, None, None, 8, 4, 9, 1, 1, 1) != 0  # kernel > 7
    assert lib.ojf_segconv_create(ctypes.byref(h), w.ctypes.data, None, None, 8, 4, 3, 0, 1, 1) != 0  # stride 0
    bad = w.copy()
    bad[1, 2, 0, 1] = np.nan
    assert lib.ojf_segconv_create(ctypes.byref(h), bad.ctypes.data, None, None, 8, 4, 3, 1, 1, 1) != 0
    assert b'non-finite' in lib.ojf_last_error()
    scale = np.array([1, np.inf, 1, 1], dtype=np.float32)
    assert lib.ojf_segconv_create(ctypes.byref(h), w.ctypes.data, scale.ctypes.data, None, 8, 4, 3, 1, 1, 1) != 0
    assert lib.ojf_segdeconv_create(ctypes.byref(h), w.ctypes.data, None, None, 4, 8, 3) != 0  # odd stride
    assert lib.ojf_segdeconv_create(ctypes.byref(h), None, None, None, 4, 8, 2) != 0
    assert h.value is None
    assert lib.ojf_segconv_forward(None, None, 8, None, 8, None, 0, None, 0, 0, 4, 4, None) != 0
    lib.ojf_segconv_destroy(None)  # a no-op, like free(NULL)
    assert lib.ojf_mesh_workspace_bytes(1, 4, 4) == 0 and lib.ojf_mesh_workspace_bytes(65, 5, 3) == 4 * 1 * 1 * 64
    assert lib.ojf_mesh_extract(None, None, None, 4, 4, 4, 0.0, None, 1.0, None, 0, None, None, None, 0, None, None) != 0
    assert lib.ojf_points_within(None, 0, None, None, None, 1.0, 1, 1, 1, 0.5, None, None, None) != 0


def test_many_scene_entry_points_validate_without_a_device():
    """ojf_extract_many / ojf_integrate_many (round 6): job count and null jobs are refused before any HIP call; the ctypes job
    structures have the C layout (sizes as a C compiler lays them out on x86-64)."""
    import ctypes
    lib = _lib.load()
    assert ctypes.sizeof(_lib.ExtractJob) == 88 and ctypes.sizeof(_lib.IntegrateJob) == 128
    assert lib.ojf_extract_many(0, None, 8, 8, 8, 4, 4, 9, -0.1, None) != 0
    assert lib.ojf_extract_many(_lib.MAX_SCENES + 1, (_lib.ExtractJob * 9)(), 8, 8, 8, 4, 4, 9, -0.1, None) != 0
    jobs = (_lib.ExtractJob * 2)()
    assert lib.ojf_extract_many(2, jobs, 8, 8, 8, 4, 4, 9, -0.1, None) != 0 and b'null' in lib.ojf_last_error()
    assert lib.ojf_extract_many(2, jobs, 8, 8, 8, 4, 4, 8, -0.1, None) != 0 and b'odd' in lib.ojf_last_error()
    ij = (_lib.IntegrateJob * 2)()
    assert lib.ojf_integrate_many(2, ij, 9, 7, 0.1, 8, 8, 8, 4, 4, None) != 0 and b'null' in lib.ojf_last_error()
    assert lib.ojf_integrate_many(0, ij, 9, 7, 0.1, 8, 8, 8, 4, 4, None) != 0
    assert lib.ojf_integrate_many(2, ij, 9, 11, 0.1, 8, 8, 8, 4, 4, None) != 0 and b'n_tail' in lib.ojf_last_error()
