"""perception_amd - MI355X-native implementation of the cuboid_detection point-cloud path.

The compute path is libcuboid_hip.so (hand-written HIP for gfx950 behind the C-ABI of
include/cuboid_hip.h).  The Python here is host plumbing only: the ctypes binding
(capi), PCD I/O (pcd), the template generator (templates), the synthetic D435 frame
generator (synth) and the frame-per-GPU batch driver (batch).
"""
__all__ = ["capi", "pcd", "templates", "synth", "batch"]
