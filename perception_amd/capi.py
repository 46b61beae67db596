"""ctypes binding of libcuboid_hip.so (the C-ABI declared in include/cuboid_hip.h).

This is the only way Python reaches the HIP path; there is no CPU fallback.  Loading
fails loudly when the shared library has not been built (run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C perception_amd/csrc`).
"""
import ctypes as C
import os

import numpy as np

CD_ABI_VERSION = 4
CD_MAX_TEMPLATES = 8
CD_MAX_CLUSTERS_PER_FRAME = 8
CD_FRAME_MORE_CLUSTERS = 1

CD_OK = 0
CD_ERR_INVALID_ARG = -1
CD_ERR_CAPACITY = -2
CD_ERR_DEVICE = -3
CD_ERR_NO_MODEL = -4
CD_ERR_FEW_CORRESPONDENCES = -5
CD_ERR_LEAF_TOO_SMALL = -6
CD_ERR_NO_TEMPLATE = -7

LIB_PATH = os.environ.get("CUBOID_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcuboid_hip.so")

# every symbol include/cuboid_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "cd_default_params", "cd_abi_version", "cd_struct_size", "cd_create", "cd_destroy", "cd_last_error",
    "cd_set_template", "cd_crop_voxel", "cd_segment_plane", "cd_extract", "cd_surface_frame", "cd_bbox_filter", "cd_cluster", "cd_icp",
    "cd_process_batch", "cd_process_frame", "cd_process_batch_device", "cd_get_cluster_results", "cd_pose_to_position_quaternion",
    "cd_bbox_corners", "cd_get_timing", "cd_get_frame_cloud", "cd_get_cluster_points", "cd_ground_plane", "cd_set_frame_guesses",
    "cd_template_lattice_faces", "cd_template_nearest", "cd_lattice_detect", "cd_passthrough",
]

CD_CLOUD_VOXELS, CD_CLOUD_OBJECTS = 0, 1
CD_GUESS_NONE, CD_GUESS_PARAMS, CD_GUESS_PER_FRAME = 0, 1, 2


CD_PLANE, CD_PLANE_PERPENDICULAR, CD_PLANE_PARALLEL = 0, 1, 2


class CdSurfaceFrameResult(C.Structure):
    _fields_ = [("Rt", C.c_float * 16), ("coeff", (C.c_float * 4) * 3), ("midpoint", (C.c_float * 4) * 3),
                ("n_points", C.c_int32 * 3), ("iterations", C.c_int32 * 3), ("reserved", C.c_int32 * 2)]


class CdParams(C.Structure):
    _fields_ = [
        ("crop_z_min", C.c_double), ("crop_z_max", C.c_double),
        ("crop_x_min", C.c_double), ("crop_x_max", C.c_double),
        ("leaf_size", C.c_float), ("rgb_offset", C.c_int32),
        ("plane_distance_threshold", C.c_double),
        ("plane_max_iterations", C.c_int32), ("plane_optimize", C.c_int32),
        ("plane_probability", C.c_double),
        ("extract_negative", C.c_int32), ("crop2_enable", C.c_int32),
        ("crop2_z_min", C.c_double), ("crop2_z_max", C.c_double),
        ("cluster_enable", C.c_int32),
        ("cluster_min_size", C.c_int32), ("cluster_max_size", C.c_int32),
        ("cluster_tolerance", C.c_double),
        ("icp_max_iterations", C.c_int32), ("template_slot", C.c_int32),
        ("icp_transformation_epsilon", C.c_double),
        ("icp_euclidean_fitness_epsilon", C.c_double),
        ("icp_accept_fitness", C.c_double),
        ("bbox_P", C.c_double * 12), ("bbox_enable", C.c_int32), ("bbox_rect", C.c_int32 * 4),
        ("plane_model", C.c_int32), ("plane_axis", C.c_float * 3), ("plane_eps_angle", C.c_double),
        ("icp_use_guess", C.c_int32), ("icp_guess", C.c_float * 16),
    ]


class CdClusterResult(C.Structure):
    _fields_ = [
        ("size", C.c_int32), ("iterations", C.c_int32),
        ("converged", C.c_int32), ("accepted", C.c_int32),
        ("template_slot", C.c_int32), ("reserved", C.c_int32),
        ("T", C.c_float * 16), ("fitness", C.c_double), ("pose", C.c_double * 16),
    ]


class CdFrameResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("n_cropped", C.c_int32), ("n_voxels", C.c_int32),
        ("n_plane", C.c_int32), ("n_objects", C.c_int32), ("n_clusters", C.c_int32),
        ("ransac_iterations", C.c_int32), ("flags", C.c_int32),
        ("plane", C.c_float * 4), ("pad", C.c_float * 4),
        ("clusters", CdClusterResult * CD_MAX_CLUSTERS_PER_FRAME),
    ]


class CdTiming(C.Structure):
    _fields_ = [
        ("stage_ms", C.c_float * 5), ("icp_kernel_ms", C.c_float),
        ("icp_kernel_launches", C.c_int32),
        ("icp_pair_tests_lo", C.c_int32), ("icp_pair_tests_hi", C.c_int32), ("icp_persist_gave_up", C.c_int32),
        ("algorithmic_bytes", C.c_int64), ("icp_algorithmic_bytes", C.c_int64),
        ("scan_retries", C.c_int32), ("icp_regime", C.c_int32),
        ("icp_handovers", C.c_int32), ("icp_search", C.c_int32),
        ("icp_handover_lost", C.c_int32), ("icp_wave_ms", C.c_float),
    ]


FRAME_RESULT_BYTES = C.sizeof(CdFrameResult)


def default_params():
    """cuboid_detection launch values (ground_plane_segmentation.launch:14-18,
    iterative_closest_point.launch:39-42) + object_detection's cluster constants
    (object_pose_detection.cpp:356-358).  Pure Python mirror of cd_default_params()."""
    p = CdParams()
    p.crop_z_min, p.crop_z_max = 0.0, 0.9
    p.crop_x_min, p.crop_x_max = -0.2, 0.2
    p.leaf_size = 0.005
    p.rgb_offset = -1
    p.plane_distance_threshold = 0.015
    p.plane_max_iterations = 1000
    p.plane_optimize = 1
    p.plane_probability = 0.99
    p.extract_negative = 1
    p.crop2_enable = 1
    p.crop2_z_min, p.crop2_z_max = 0.0, 0.75
    p.cluster_enable = 1
    p.cluster_min_size, p.cluster_max_size = 200, 25000
    p.cluster_tolerance = 0.02
    p.icp_max_iterations = 5000
    p.template_slot = 0
    p.icp_transformation_epsilon = 1e-9
    p.icp_euclidean_fitness_epsilon = 0.0004
    p.icp_accept_fitness = 0.0004
    return p


_lib = None


def load_library(path=None):
    """dlopen libcuboid_hip.so and set prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "libcuboid_hip.so not found at %s: the HIP extension is not built and there is "
            "no CPU fallback (build with `make -C perception_amd/csrc`)" % p)
    lib = C.CDLL(p)
    vp, i32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    lib.cd_default_params.argtypes = [C.POINTER(CdParams)]
    lib.cd_default_params.restype = None
    lib.cd_abi_version.restype = C.c_int
    lib.cd_struct_size.argtypes = [C.c_int]
    lib.cd_struct_size.restype = C.c_int
    lib.cd_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.cd_destroy.argtypes = [vp]
    lib.cd_destroy.restype = None
    lib.cd_last_error.argtypes = [vp]
    lib.cd_last_error.restype = C.c_char_p
    lib.cd_set_template.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int]
    lib.cd_crop_voxel.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(CdParams), vp, vp,
                                  C.c_int, ip, ip]
    lib.cd_segment_plane.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(CdParams), vp, vp,
                                     C.c_int, ip, ip]
    lib.cd_surface_frame.argtypes = [vp, vp, C.c_size_t, C.c_int, f32p, C.c_int, C.POINTER(CdParams), C.POINTER(CdSurfaceFrameResult)]
    lib.cd_bbox_filter.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp, vp, C.c_int, ip]
    lib.cd_extract.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, ip]
    lib.cd_cluster.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(CdParams), vp, vp, C.c_int, ip]
    lib.cd_icp.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(CdParams),
                           C.POINTER(CdClusterResult), vp]
    for f in (lib.cd_process_batch, lib.cd_process_batch_device):
        f.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(CdParams), vp, vp, vp]
    lib.cd_process_frame.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(CdParams), vp, vp, vp]
    lib.cd_get_cluster_results.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(CdClusterResult), ip]
    lib.cd_pose_to_position_quaternion.argtypes = [C.POINTER(C.c_double)] * 3
    lib.cd_pose_to_position_quaternion.restype = None
    lib.cd_bbox_corners.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, f32p]
    lib.cd_bbox_corners.restype = None
    lib.cd_get_timing.argtypes = [vp, C.POINTER(CdTiming)]
    lib.cd_get_frame_cloud.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, ip]
    lib.cd_get_cluster_points.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, ip]
    lib.cd_ground_plane.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(CdParams), f32p, vp, C.c_int, ip, ip]
    lib.cd_set_frame_guesses.argtypes = [vp, f32p, C.c_int]
    lib.cd_template_lattice_faces.argtypes = [vp, C.c_int]
    lib.cd_template_nearest.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int, vp, vp]
    lib.cd_lattice_detect.argtypes = [vp, C.c_size_t, C.c_int, vp]
    lib.cd_passthrough.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, C.c_int, ip]
    if path is None:
        _lib = lib
    return lib


class CuboidError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("cd status %d: %s" % (status, msg))
        self.status = status


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _points(a):
    """(base pointer, stride, n) of a 2-D float32 array whose rows start with x,y,z."""
    a = np.ascontiguousarray(a, dtype=np.float32) if a.dtype != np.float32 or not a.flags.c_contiguous else a
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, (a.strides[0] if a.shape[0] else 4 * a.shape[1]), a.shape[0]   # numpy reports stride 0 for empty arrays


class Context:
    """One GPU context (cd_create/cd_destroy).  Not thread-safe, like the reference node."""

    def __init__(self, max_points, max_frames=1, device_id=0):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.cd_create(device_id, int(max_points), int(max_frames), C.byref(h))
        if st != CD_OK:
            raise CuboidError(st, "cd_create failed (no usable HIP device? there is no CPU fallback)")
        self.h = h
        self.max_points, self.max_frames = int(max_points), int(max_frames)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cd_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, st, ok=(CD_OK,)):
        if st not in ok:
            raise CuboidError(st, (self.lib.cd_last_error(self.h) or b"").decode())
        return st

    def set_template(self, slot, xyz):
        a, stride, m = _points(xyz)
        self._check(self.lib.cd_set_template(self.h, slot, _ptr(a), stride, m))

    def crop_voxel(self, points, prm, want_rgb=False):
        a, stride, n = _points(points)
        out = np.empty((n, 3), np.float32)
        rgb = np.empty(n, np.uint32) if want_rgb else None
        nc, nv = C.c_int(), C.c_int()
        self._check(self.lib.cd_crop_voxel(self.h, _ptr(a), stride, n, C.byref(prm), _ptr(out), _ptr(rgb),
                                           n, C.byref(nc), C.byref(nv)))
        return out[:nv.value].copy(), (rgb[:nv.value].copy() if want_rgb else None), nc.value

    def segment_plane(self, xyz, prm):
        a, stride, n = _points(xyz)
        coeff = np.zeros(4, np.float32)
        inl = np.empty(max(n, 1), np.int32)
        ni, it = C.c_int(), C.c_int()
        st = self.lib.cd_segment_plane(self.h, _ptr(a), stride, n, C.byref(prm), _ptr(coeff), _ptr(inl), n,
                                       C.byref(ni), C.byref(it))
        self._check(st, ok=(CD_OK, CD_ERR_NO_MODEL))
        return st, coeff, inl[:ni.value].copy(), it.value

    def cluster(self, xyz, prm, sizes_capacity=4096):
        a, stride, n = _points(xyz)
        labels = np.empty(max(n, 1), np.int32)
        sizes = np.zeros(sizes_capacity, np.int32)
        k = C.c_int()
        self._check(self.lib.cd_cluster(self.h, _ptr(a), stride, n, C.byref(prm), _ptr(labels), _ptr(sizes),
                                        sizes_capacity, C.byref(k)))
        return labels[:n].copy(), sizes[:min(k.value, sizes_capacity)].copy(), k.value

    def surface_frame(self, xyz, table_normal, prm, invert=True):
        """surface_normal_estimation.cpp's callback: three axis-constrained plane fits -> pose of the cuboid frame."""
        a, stride, n = _points(xyz)
        tn = np.ascontiguousarray(table_normal, np.float32)
        res = CdSurfaceFrameResult()
        st = self.lib.cd_surface_frame(self.h, _ptr(a), stride, n, tn.ctypes.data_as(C.POINTER(C.c_float)), 1 if invert else 0,
                                       C.byref(prm), C.byref(res))
        self._check(st, ok=(CD_OK, CD_ERR_NO_MODEL))
        return st, res

    def bbox_filter(self, xyz, P, rect):
        """bbox_filter.cpp: ascending indices of the points projecting strictly inside the rectangle."""
        a, stride, n = _points(xyz)
        Pm = np.ascontiguousarray(P, np.float64).ravel()
        r = np.ascontiguousarray(rect, np.int32)
        idx = np.empty(max(n, 1), np.int32)
        cnt = C.c_int()
        self._check(self.lib.cd_bbox_filter(self.h, _ptr(a), stride, n, _ptr(Pm), _ptr(r), _ptr(idx), max(n, 1), C.byref(cnt)))
        return idx[:cnt.value].copy()

    def icp(self, slot, src, prm, want_aligned=False):
        a, stride, n = _points(src)
        res = CdClusterResult()
        al = np.empty((n, 3), np.float32) if want_aligned else None
        st = self.lib.cd_icp(self.h, slot, _ptr(a), stride, n, C.byref(prm), C.byref(res), _ptr(al))
        self._check(st, ok=(CD_OK, CD_ERR_FEW_CORRESPONDENCES))
        return st, res, al

    def process_batch(self, frames, prm, want_indices=False):
        """frames: (F, N, C>=3) float32 host array.  Returns (results, plane_inliers, labels)."""
        f = np.ascontiguousarray(frames, dtype=np.float32)
        assert f.ndim == 3
        F, N, Cc = f.shape
        res = (CdFrameResult * F)()
        pi = np.empty((F, N), np.int32) if want_indices else None
        lb = np.empty((F, N), np.int32) if want_indices else None
        self._check(self.lib.cd_process_batch(self.h, _ptr(f), Cc * 4, N, F, C.byref(prm),
                                              C.cast(res, C.c_void_p), _ptr(pi), _ptr(lb)))
        return res, pi, lb

    def extract(self, records, indices, negative=True):
        """pcl::ExtractIndices on whole records: `records` is an (n, k) array of 4-byte items (all fields of a point)."""
        r = np.ascontiguousarray(records)
        assert r.ndim == 2 and r.dtype.itemsize == 4
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        n, stride = r.shape[0], r.shape[1] * 4
        out = np.empty((n if negative else len(idx), r.shape[1]), r.dtype)
        cnt = C.c_int()
        self._check(self.lib.cd_extract(self.h, _ptr(r), stride, n, _ptr(idx), len(idx), 1 if negative else 0, _ptr(out), len(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def process_frame(self, points, prm, want_indices=False):
        """One frame (N, C>=3) float32: the reference's callback body as one call.  Returns (result, plane_inliers, labels)."""
        f = np.ascontiguousarray(points, dtype=np.float32)
        assert f.ndim == 2
        N, Cc = f.shape
        res = CdFrameResult()
        pi = np.empty(N, np.int32) if want_indices else None
        lb = np.empty(N, np.int32) if want_indices else None
        self._check(self.lib.cd_process_frame(self.h, _ptr(f), Cc * 4, N, C.byref(prm), C.byref(res), _ptr(pi), _ptr(lb)))
        return res, pi, lb

    def process_batch_device(self, dev_ptr, stride_bytes, points_per_frame, n_frames, prm,
                             results=None, plane_inliers=None, labels=None):
        """Input already in HBM (e.g. a torch tensor's data_ptr())."""
        res = results if results is not None else (CdFrameResult * n_frames)()
        self._check(self.lib.cd_process_batch_device(self.h, C.c_void_p(dev_ptr), stride_bytes,
                                                     points_per_frame, n_frames, C.byref(prm),
                                                     C.cast(res, C.c_void_p), _ptr(plane_inliers),
                                                     _ptr(labels)))
        return res

    def process_batch_host_ptr(self, host_ptr, stride_bytes, points_per_frame, n_frames, prm, results=None):
        """cd_process_batch on a raw HOST pointer (e.g. a pinned torch tensor's data_ptr()): the upload is part of the call."""
        res = results if results is not None else (CdFrameResult * n_frames)()
        self._check(self.lib.cd_process_batch(self.h, C.c_void_p(host_ptr), stride_bytes, points_per_frame, n_frames,
                                              C.byref(prm), C.cast(res, C.c_void_p), None, None))
        return res

    def cluster_results(self, frame, first=0, count=None):
        """ICP results of EVERY cluster of `frame` of the last process_batch* call (opd.cpp:376 registers all of them; the
        fixed-size record holds the CD_MAX_CLUSTERS_PER_FRAME largest)."""
        tot = C.c_int()
        self._check(min(0, self.lib.cd_get_cluster_results(self.h, frame, 0, 0, None, C.byref(tot))))
        n = max(0, tot.value - first) if count is None else count
        out = (CdClusterResult * max(n, 1))()
        got = self.lib.cd_get_cluster_results(self.h, frame, first, n, out, None)
        self._check(min(0, got))
        return [out[i] for i in range(got)]

    def frame_cloud(self, frame, which, stride_bytes=16, rgb_offset=12):
        """Voxel cloud (CD_CLOUD_VOXELS) or object cloud (CD_CLOUD_OBJECTS) of `frame` of the last process_batch* call, as
        (n, stride_bytes / 4) uint32 records: x,y,z at words 0..2, packed rgb at rgb_offset (-1: none), the rest zero."""
        n = C.c_int()
        st = self.lib.cd_get_frame_cloud(self.h, frame, which, None, stride_bytes, rgb_offset, 0, C.byref(n))
        if st not in (CD_OK, CD_ERR_CAPACITY):
            self._check(st)
        out = np.zeros((max(n.value, 1), stride_bytes // 4), np.uint32)
        self._check(self.lib.cd_get_frame_cloud(self.h, frame, which, _ptr(out), stride_bytes, rgb_offset, n.value, C.byref(n)))
        return out[:n.value]

    def cluster_points(self, frame, k, aligned=False, stride_bytes=16):
        """Points of cluster k of `frame` of the last process_batch* call (aligned: the cloud icp.align returned), as
        (n, stride_bytes / 4) float32 records (fourth word 1.0f: pcl::PointXYZ's wire layout)."""
        n = C.c_int()
        st = self.lib.cd_get_cluster_points(self.h, frame, k, 1 if aligned else 0, None, stride_bytes, 0, C.byref(n))
        if st not in (CD_OK, CD_ERR_CAPACITY):
            self._check(st)
        out = np.zeros((max(n.value, 1), stride_bytes // 4), np.float32)
        self._check(self.lib.cd_get_cluster_points(self.h, frame, k, 1 if aligned else 0, _ptr(out), stride_bytes, n.value, C.byref(n)))
        return out[:n.value]

    def ground_plane(self, points, prm):
        """gps.cpp:43-112 as one call.  points: (n, k) float32 records.  Returns (status, coeff, kept records (m, k) uint32 view
        of the input's layout, n_inliers)."""
        a = np.ascontiguousarray(points, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] >= 3
        n, stride = a.shape[0], a.shape[1] * 4
        coeff = np.zeros(4, np.float32)
        out = np.zeros((max(n, 1), a.shape[1]), np.uint32)
        m, ni = C.c_int(), C.c_int()
        st = self.lib.cd_ground_plane(self.h, _ptr(a), stride, n, C.byref(prm), coeff.ctypes.data_as(C.POINTER(C.c_float)), _ptr(out), n,
                                      C.byref(m), C.byref(ni))
        self._check(st, ok=(CD_OK, CD_ERR_NO_MODEL))
        return st, coeff, out[:m.value].copy(), ni.value

    def set_frame_guesses(self, guesses):
        """Per-frame initial guesses (F, 4, 4) float32 for prm.icp_use_guess = CD_GUESS_PER_FRAME; None clears."""
        if guesses is None:
            self._check(self.lib.cd_set_frame_guesses(self.h, None, 0))
            return
        g = np.ascontiguousarray(guesses, np.float32).reshape(-1, 16)
        self._check(self.lib.cd_set_frame_guesses(self.h, g.ctypes.data_as(C.POINTER(C.c_float)), g.shape[0]))

    def passthrough(self, records, field, lo, hi, negative=False):
        """pcl::PassThrough on whole records: `records` (n, k >= 3) of 4-byte items; field 'x' / 'y' / 'z' or None."""
        r = np.ascontiguousarray(records)
        assert r.ndim == 2 and r.dtype.itemsize == 4 and r.shape[1] >= 3
        out = np.empty_like(r)
        cnt = C.c_int()
        f = -1 if field is None else "xyz".index(field)
        self._check(self.lib.cd_passthrough(self.h, _ptr(r), r.shape[1] * 4, r.shape[0], f, float(lo), float(hi), 1 if negative else 0,
                                            _ptr(out), r.shape[0], C.byref(cnt)))
        return out[:cnt.value].copy()

    def template_lattice_faces(self, slot):
        """Faces of the slot's template as a union of axis-aligned lattices (make_cuboid.py's output); 0 = an arbitrary cloud."""
        return self._check(self.lib.cd_template_lattice_faces(self.h, slot), ok=tuple(range(0, 9)))

    def template_nearest(self, slot, queries):
        """Nearest template point of every query: (original index int32, squared distance float32).  Lattice templates only."""
        a, stride, n = _points(queries)
        idx = np.empty(max(n, 1), np.int32)
        d2 = np.empty(max(n, 1), np.float32)
        self._check(self.lib.cd_template_nearest(self.h, slot, _ptr(a), stride, n, _ptr(idx), _ptr(d2)))
        return idx[:n], d2[:n]

    def timing(self):
        t = CdTiming()
        self._check(self.lib.cd_get_timing(self.h, C.byref(t)))
        return t


def lattice_detect(xyz):
    """Host-only lattice test of cd_set_template: list of faces (constant axis, fast axis, first index, n_fast, n_slow)."""
    lib = load_library()
    a, stride, m = _points(xyz)
    out = np.zeros((8, 5), np.int32)
    nf = lib.cd_lattice_detect(_ptr(a), stride, m, _ptr(out))
    if nf < 0:
        raise CuboidError(nf, "cd_lattice_detect")
    return [tuple(int(v) for v in out[f]) for f in range(nf)]


def results_to_array(res):
    """View an array of CdFrameResult as a uint8 numpy array (F, FRAME_RESULT_BYTES)."""
    n = len(res)
    return np.frombuffer(res, dtype=np.uint8).reshape(n, FRAME_RESULT_BYTES)


def results_from_array(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    n = arr.shape[0]
    out = (CdFrameResult * n)()
    C.memmove(out, arr.ctypes.data, n * FRAME_RESULT_BYTES)
    return out
