"""CPU checks of the drop-in boundary: libcuboid_hip.so loads, exports every symbol that
include/cuboid_hip.h declares, its struct layout matches the ctypes mirror, and without a
GPU it refuses to create a context instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, rot_xyz
from perception_amd import capi


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "cuboid_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cd_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.cd_abi_version() == capi.CD_ABI_VERSION == 4


def test_struct_layout_matches_header():
    lib = capi.load_library()
    assert lib.cd_struct_size(0) == C.sizeof(capi.CdParams)
    assert lib.cd_struct_size(1) == C.sizeof(capi.CdClusterResult)
    assert lib.cd_struct_size(2) == C.sizeof(capi.CdFrameResult)
    assert lib.cd_struct_size(3) == C.sizeof(capi.CdTiming)
    p = capi.CdParams()
    lib.cd_default_params(C.byref(p))
    q = capi.default_params()
    for name, ty in capi.CdParams._fields_:
        a, b = getattr(p, name), getattr(q, name)
        if hasattr(a, "__len__"):
            a, b = list(a), list(b)
        assert a == b, name


def test_host_side_s7_helpers_match_oracle(O):
    """cd_pose_to_position_quaternion / cd_bbox_corners are host code (no GPU needed)."""
    lib = capi.load_library()
    H = np.eye(4)
    H[:3, :3] = rot_xyz(0.3, -0.7, 2.1)
    H[:3, 3] = [0.1, -0.2, 0.55]
    pos, q = np.zeros(3), np.zeros(4)
    dp = C.POINTER(C.c_double)
    lib.cd_pose_to_position_quaternion(H.ctypes.data_as(dp), pos.ctypes.data_as(dp), q.ctypes.data_as(dp))
    pos_o, q_o = O.pose_to_position_quaternion(H)
    assert np.array_equal(pos, pos_o) and np.array_equal(q, q_o)
    out = np.zeros((8, 3), np.float32)
    lib.cd_bbox_corners(H.ctypes.data_as(dp), 0.2, 0.1, 0.03, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out, O.bbox_corners(H, 0.2, 0.1, 0.03))


def test_no_gpu_means_no_context():
    """There is no CPU fallback: without a usable HIP device cd_create must fail."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.CuboidError) as e:
        capi.Context(1024, 1)
    assert e.value.status == capi.CD_ERR_DEVICE
