"""PCD v0.7 reader/writer (ASCII and binary), replacing pcl::io::loadPCDFile for the
templates (reference: cuboid_detection/src/iterative_closest_point.cpp:159,
object_detection/src/object_pose_detection.cpp:398)."""
import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16,
       ("U", 4): np.uint32, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def read_pcd(path):
    """Returns (fields dict name -> 1-D array, header dict).  DATA ascii|binary."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    names = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(names))]
    npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0].lower()
    out = {}
    if mode == "ascii":
        ncol = sum(counts)
        toks = raw[pos:].split()
        tab = np.array(toks[:npts * ncol], dtype=np.float64).reshape(npts, ncol)
        col = 0
        for nm, sz, ty, ct in zip(names, sizes, types, counts):
            out[nm] = tab[:, col].astype(_NP[(ty, sz)]) if ct == 1 else tab[:, col:col + ct].astype(_NP[(ty, sz)])
            col += ct
    elif mode == "binary":
        dt = np.dtype([(nm, _NP[(ty, sz)], (ct,)) if ct > 1 else (nm, _NP[(ty, sz)])
                       for nm, sz, ty, ct in zip(names, sizes, types, counts)])
        rec = np.frombuffer(raw, dtype=dt, count=npts, offset=pos)
        for nm in names:
            out[nm] = np.array(rec[nm])
    else:
        raise ValueError("unsupported PCD DATA mode %r (binary_compressed is not produced by the reference)" % mode)
    return out, hdr


def read_xyz(path):
    """(N,3) float32, the pcl::PointCloud<pcl::PointXYZ> view of the file."""
    f, _ = read_pcd(path)
    return np.stack([f["x"], f["y"], f["z"]], axis=1).astype(np.float32)


PCD_HEADER = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\n"
              "TYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n")


def pcd_ascii_bytes(xyz):
    """ASCII PCD bytes in the layout the reference's template files use ('%f %f %f')."""
    xyz = np.asarray(xyz, dtype=np.float64)
    body = "".join("%f %f %f\n" % (r[0], r[1], r[2]) for r in xyz)
    return (PCD_HEADER % (len(xyz), len(xyz)) + body).encode("ascii")


def write_pcd_ascii(path, xyz):
    with open(path, "wb") as f:
        f.write(pcd_ascii_bytes(xyz))
