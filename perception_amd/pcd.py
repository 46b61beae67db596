"""PCD v0.7 reader/writer (ASCII, binary and binary_compressed), replacing pcl::io::loadPCDFile for the
templates (reference: cuboid_detection/src/iterative_closest_point.cpp:159,
object_detection/src/object_pose_detection.cpp:398)."""
import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16,
       ("U", 4): np.uint32, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def lzf_decompress(src, out_len):
    """LZF (the codec of DATA binary_compressed): a control byte c < 32 copies c+1 literals; otherwise a back reference of
    length (c >> 5) + 2 (a length field of 7 takes one more byte) at distance ((c & 31) << 8 | next) + 1."""
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        i += 1
        if c < 32:
            out += src[i:i + c + 1]
            i += c + 1
        else:
            ln = c >> 5
            if ln == 7:
                ln += src[i]
                i += 1
            ref = len(out) - (((c & 31) << 8) | src[i]) - 1
            i += 1
            if ref < 0:
                raise ValueError("corrupt LZF stream")
            for _ in range(ln + 2):              # may overlap its own output: byte by byte
                out.append(out[ref])
                ref += 1
    if len(out) != out_len:
        raise ValueError("LZF stream decodes to %d bytes, header says %d" % (len(out), out_len))
    return bytes(out)


def read_pcd(path):
    """Returns (fields dict name -> 1-D array, header dict).  DATA ascii|binary|binary_compressed."""
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
    elif mode == "binary_compressed":
        # u32 compressed size, u32 raw size, LZF stream; the raw block is field-major (all x, then all y, ...)
        csz, usz = np.frombuffer(raw, dtype="<u4", count=2, offset=pos)
        blob = lzf_decompress(raw[pos + 8:pos + 8 + int(csz)], int(usz))
        off = 0
        for nm, sz, ty, ct in zip(names, sizes, types, counts):
            a = np.frombuffer(blob, dtype=_NP[(ty, sz)], count=npts * ct, offset=off)
            out[nm] = np.array(a if ct == 1 else a.reshape(npts, ct))
            off += npts * ct * sz
    else:
        raise ValueError("unsupported PCD DATA mode %r" % mode)
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
