"""pclhip::io::loadPCDFile (perception_amd/cpp/pcl_compat.hpp) - the reader the node shims use where the reference calls
pcl::io::loadPCDFile<pcl::PointXYZ> (cuboid_detection/src/iterative_closest_point.cpp:159,
object_detection/src/object_pose_detection.cpp:398): the reference's own template files, and the PCD v0.7 header cases a
user template can bring - fields of COUNT > 1, double and integer coordinates, no POINTS line, all three DATA encodings.
Host only (g++), no GPU, no library."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from perception_amd import pcd

CPP = os.path.join(ROOT, "perception_amd", "cpp")


@pytest.fixture(scope="module")
def tool():
    subprocess.run(["make", "-C", CPP, "pcd_tool"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "pcd_tool")


def _load(tool, path, tmp_path):
    out = str(tmp_path / "out.bin")
    r = subprocess.run([tool, str(path), out], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r
    t = r.stdout.split()
    return np.fromfile(out, np.float32).reshape(-1, 3), dict(points=int(t[1]), width=int(t[3]), height=int(t[5]))


def _header(fields, sizes, types, counts, n, data, width=None, height=None, points=True):
    h = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n" % (
        " ".join(fields), " ".join(map(str, sizes)), " ".join(types), " ".join(map(str, counts)))
    h += "WIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\n" % (n if width is None else width, 1 if height is None else height)
    if points:
        h += "POINTS %d\n" % n
    return (h + "DATA %s\n" % data).encode()


@pytest.mark.parametrize("name", ["marker", "screwdriver", "eraser", "clamp"])
def test_reference_object_templates(tool, tmp_path, name):
    """the four files object_pose_detection loads (opd.cpp:87-88): same points as the Python reader, bit for bit"""
    for suffix in ("_ascii.pcd", "_ascii_tf.pcd"):
        path = os.path.join(GOLDEN, name + suffix)
        got, info = _load(tool, path, tmp_path)
        want = pcd.read_xyz(path).astype(np.float32)
        assert got is not None and info["points"] == len(want)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cuboid_templates(tool, tmp_path):
    for fn in ("template_cuboid_L200_W100_H75.pcd", "template_cuboid_L200_W75_H100_3faces.pcd"):
        got, info = _load(tool, os.path.join(GOLDEN, fn), tmp_path)
        want = pcd.read_xyz(os.path.join(GOLDEN, fn)).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_binary_with_count_and_mixed_types(tool, tmp_path):
    """x y z rgb normal[COUNT 3] curvature as PCL writes PointXYZRGBNormal-like clouds: the record stride must include the
    COUNT-3 field; a uint8 field before x shifts every offset"""
    rng = np.random.RandomState(5)
    n = 257
    xyz = rng.randn(n, 3).astype(np.float32)
    dt = np.dtype([("flag", "u1"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<f4"), ("normal", "<f4", 3), ("curv", "<f4")])
    rec = np.zeros(n, dt)
    rec["flag"] = 7
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["normal"] = rng.randn(n, 3)
    rec["rgb"] = 1.5
    p = tmp_path / "b.pcd"
    p.write_bytes(_header(["flag", "x", "y", "z", "rgb", "normal", "curv"], [1, 4, 4, 4, 4, 4, 4], list("UFFFFFF"), [1, 1, 1, 1, 1, 3, 1], n, "binary") + rec.tobytes())
    got, info = _load(tool, p, tmp_path)
    assert np.array_equal(got.view(np.uint32), xyz.view(np.uint32)) and info["points"] == n


def test_ascii_with_count_and_nan_and_no_points_line(tool, tmp_path):
    """ascii rows carry COUNT tokens per field; POINTS is optional (WIDTH x HEIGHT); nan survives"""
    rows = ["0.5 9 9 9 1 2 3", "0.25 8 8 8 -1 nan 3.5", "1 7 7 7 0 0 1e-3", "2 6 6 6 4 5 6"]
    p = tmp_path / "a.pcd"
    p.write_bytes(_header(["intensity", "normal", "x", "y", "z"], [4, 4, 4, 4, 4], list("FFFFF"), [1, 3, 1, 1, 1], 4, "ascii", width=2, height=2, points=False)
                  + ("\n".join(rows) + "\n").encode())
    got, info = _load(tool, p, tmp_path)
    want = np.array([[1, 2, 3], [-1, np.nan, 3.5], [0, 0, 1e-3], [4, 5, 6]], np.float32)
    assert info == dict(points=4, width=2, height=2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_double_and_integer_coordinates(tool, tmp_path):
    n = 33
    rng = np.random.RandomState(6)
    xd = rng.randn(n, 3)
    dt = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("k", "<i2")])
    rec = np.zeros(n, dt)
    rec["x"], rec["y"], rec["z"] = xd[:, 0], xd[:, 1], xd[:, 2]
    p = tmp_path / "d.pcd"
    p.write_bytes(_header(["x", "y", "z", "k"], [8, 8, 8, 2], list("FFFI"), [1, 1, 1, 1], n, "binary") + rec.tobytes())
    got, _ = _load(tool, p, tmp_path)
    assert np.array_equal(got, xd.astype(np.float32))
    dt = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4")])
    rec = np.zeros(3, dt)
    rec["x"], rec["y"], rec["z"] = [1, -2, 3], [4, 5, -6], [7, 8, 9]
    p.write_bytes(_header(["x", "y", "z"], [4, 4, 4], list("III"), [1, 1, 1], 3, "binary") + rec.tobytes())
    got, _ = _load(tool, p, tmp_path)
    assert np.array_equal(got, np.array([[1, 4, 7], [-2, 5, 8], [3, -6, 9]], np.float32))


def _lzf_literals(raw):
    """a valid LZF stream made of literal runs only"""
    out = bytearray()
    for i in range(0, len(raw), 32):
        chunk = raw[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def test_binary_compressed(tool, tmp_path):
    """DATA binary_compressed: sizes + LZF stream of the fields stored one after the other.  One stream of literal runs,
    one with back references (a long run of equal bytes: length-7+ reference with an extension byte, overlapping copy)."""
    n = 50
    rng = np.random.RandomState(8)
    xyz = rng.randn(n, 3).astype(np.float32)
    rgb = np.full(n, 2.5, np.float32)
    soa = xyz[:, 0].tobytes() + xyz[:, 1].tobytes() + xyz[:, 2].tobytes() + rgb.tobytes()
    comp = _lzf_literals(soa)
    p = tmp_path / "c.pcd"
    p.write_bytes(_header(["x", "y", "z", "rgb"], [4, 4, 4, 4], list("FFFF"), [1, 1, 1, 1], n, "binary_compressed") + struct.pack("<II", len(comp), len(soa)) + comp)
    got, _ = _load(tool, p, tmp_path)
    assert np.array_equal(got.view(np.uint32), xyz.view(np.uint32))
    assert np.array_equal(pcd.read_xyz(str(p)).view(np.uint32), xyz.view(np.uint32))      # the Python reader agrees
    # all-zero z and rgb columns written as ONE back reference chain: literal 0x00, then references of distance 1
    zeros = 2 * 4 * n                      # bytes of the z and rgb columns
    xy = xyz[:, 0].tobytes() + xyz[:, 1].tobytes()
    comp = bytearray(_lzf_literals(xy))
    comp += bytes([0, 0])                  # literal run of one zero byte
    left = zeros - 1
    while left > 0:
        ln = min(left, 264)                # len field 7 + extension byte 255 -> 262 + 2
        if ln >= 9:
            comp += bytes([(7 << 5) | 0, ln - 9, 0])          # distance 1: high bits 0, low byte 0
        else:
            ln = max(ln, 3) if left >= 3 else left
            if ln < 3:                     # too short for a reference: literals
                comp += bytes([ln - 1]) + bytes(ln)
            else:
                comp += bytes([((ln - 2) << 5) | 0, 0])
        left -= ln
    soa2 = xy + bytes(zeros)
    p.write_bytes(_header(["x", "y", "z", "rgb"], [4, 4, 4, 4], list("FFFF"), [1, 1, 1, 1], n, "binary_compressed") + struct.pack("<II", len(comp), len(soa2)) + bytes(comp))
    got, _ = _load(tool, p, tmp_path)
    want = xyz.copy()
    want[:, 2] = 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(pcd.read_xyz(str(p)).view(np.uint32), want.view(np.uint32))


def test_errors_are_reported(tool, tmp_path):
    p = tmp_path / "e.pcd"
    for blob, msg in ((_header(["x", "y"], [4, 4], list("FF"), [1, 1], 1, "ascii") + b"1 2\n", "no x / y / z"),
                      (_header(["x", "y", "z"], [4, 4, 4], list("FFF"), [1, 1, 1], 5, "binary") + b"\0" * 24, "ends early"),
                      (_header(["x", "y", "z"], [4, 4, 4], list("FFF"), [1, 1, 1], 2, "binary_compressed") + struct.pack("<II", 3, 24) + b"\xe0\x01\x02", "malformed"),
                      (b"VERSION 0.7\nFIELDS x y z\n", "no DATA")):
        p.write_bytes(blob)
        got, r = _load(tool, p, tmp_path)
        assert got is None and r.returncode == 1 and msg in r.stderr
    got, r = _load(tool, tmp_path / "missing.pcd", tmp_path)
    assert got is None and "cannot open" in r.stderr


def test_hostile_headers_return_minus_one_instead_of_throwing(tool, tmp_path):
    """The header is not trusted: point counts far beyond what the file holds, WIDTH x HEIGHT that overflows,
    a binary_compressed block that claims 4 GiB - every one is the documented -1 (exit code 1 of pcd_tool, a message on
    stderr), decided before anything is allocated; none ends in bad_alloc / length_error (which would abort: exit code < 0)."""
    p = tmp_path / "h.pcd"
    xyz = (["x", "y", "z"], [4, 4, 4], list("FFF"), [1, 1, 1])
    cases = (
        (_header(*xyz, 2 ** 40, "binary") + b"\0" * 24, "out of range"),
        (_header(*xyz, 2 ** 30, "binary") + b"\0" * 24, "ends early"),
        (_header(*xyz, 2 ** 30, "ascii") + b"1 2 3\n", "ends early"),
        (_header(*xyz, 0, "ascii", width=2 ** 31, height=2 ** 31, points=False) + b"1 2 3\n", "ends early"),
        (_header(*xyz, 0, "ascii", width=2 ** 62, height=4, points=False) + b"1 2 3\n", "out of range"),
        (_header(*xyz, (2 ** 32 - 4) // 12, "binary_compressed") + struct.pack("<II", 2 ** 32 - 8, 2 ** 32 - 4) + b"\0" * 16, "ends early"),
        (_header(*xyz, 2 ** 20, "binary_compressed") + struct.pack("<II", 3, 12 * 2 ** 20) + b"\xe0\x01\x02", "malformed"),
        (_header(["x", "y", "z"], [2 ** 30, 4, 4], list("FFF"), [1, 1, 1], 4, "binary") + b"\0" * 64, "bad record size"),
    )
    for blob, msg in cases:
        p.write_bytes(blob)
        got, r = _load(tool, p, tmp_path)
        assert got is None and r.returncode == 1 and msg in r.stderr, (msg, r.returncode, r.stderr[-300:])
