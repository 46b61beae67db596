import os
import sys

import numpy as np
import pytest

# The PyTorch-ROCm wheel bundles its own HIP runtime with the same SONAME (libamdhip64.so.7) as the system one that
# libcuboid_hip.so links: a process gets ONE copy, whichever is loaded first, and torch cannot initialise on the system
# copy ("No HIP GPUs are available").  Tests that use both (device-resident inputs, RCCL) need torch's copy: load it first.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _no_hip_error_left_behind(request):
    """After every GPU test: the library must not leave a failed HIP call unreported (hipGetLastError is sticky per host
    thread; an error that libcuboid_hip.so ignored would surface later in some unrelated torch call)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so.7")
    except OSError:
        return
    hip.hipGetErrorString.restype = ctypes.c_char_p
    err = hip.hipGetLastError()
    assert err == 0, "a HIP call failed silently during this test: %d (%s)" % (err, hip.hipGetErrorString(err).decode())


@pytest.fixture(scope="session")
def O():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture()
def prm():
    from perception_amd import capi
    return capi.default_params()


@pytest.fixture(scope="session")
def template():
    from perception_amd import templates
    return templates.template_xyz32(**templates.DEFAULT_TEMPLATE)


@pytest.fixture(scope="session")
def frames4():
    from perception_amd import synth
    return [synth.frame(i) for i in range(4)]


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def quat_to_matrix(x, y, z, w):
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
