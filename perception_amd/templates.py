"""Template cuboid generator: the three visible faces of an L x W x H cuboid centred at
the origin, sampled every `density` metres.  Host-side mirror of the reference's
cuboid_detection/templates/make_cuboid.py (grids :38-40, face order :53-55, text format
:66); the generated files are byte-identical to the reference's committed templates
(tests/test_oracle_golden.py pins the SHA-256)."""
import numpy as np

from . import pcd


def _grid(a, b):
    """All (a_i, b_j) pairs, first argument varying fastest."""
    aa, bb = np.meshgrid(a, b)
    return aa.ravel(), bb.ravel()


def make_cuboid_template(length=0.2, width=0.1, height=0.075, density=0.002):
    """(M,3) float64: face z=-H/2 over X x Y, face y=-W/2 over X x Z, face x=-L/2 over Y x Z."""
    X = np.arange(-length / 2.0, length / 2.0, density)
    Y = np.arange(-width / 2.0, width / 2.0, density)
    Z = np.arange(-height / 2.0, height / 2.0, density)
    fx, fy = _grid(X, Y)
    bottom = np.stack([fx, fy, np.full(fx.shape, -height / 2.0)], axis=1)
    gx, gz = _grid(X, Z)
    side = np.stack([gx, np.full(gx.shape, -width / 2.0), gz], axis=1)
    hy, hz = _grid(Y, Z)
    end = np.stack([np.full(hy.shape, -length / 2.0), hy, hz], axis=1)
    return np.concatenate([bottom, side, end], axis=0)


def template_filename(length, width, height):
    return "template_cuboid_L%d_W%d_H%d_3faces.pcd" % (length * 1000, width * 1000, height * 1000)


def template_pcd_bytes(length, width, height, density):
    return pcd.pcd_ascii_bytes(make_cuboid_template(length, width, height, density))


def template_xyz32(length, width, height, density):
    """What pcl::io::loadPCDFile<pcl::PointXYZ> yields from the written file: the text
    values (6 decimals) parsed to float32."""
    txt = template_pcd_bytes(length, width, height, density).split(b"DATA ascii\n", 1)[1]
    return np.array(txt.split(), dtype=np.float64).reshape(-1, 3).astype(np.float32)


# launch default: iterative_closest_point.launch:34,39-41
DEFAULT_TEMPLATE = dict(length=0.2, width=0.1, height=0.03, density=0.002)


def main(argv=None):
    """`python -m perception_amd.templates -L 0.2 -W 0.1 -H 0.03 -d 0.002 [-f name]`: the command line of the
    reference's make_cuboid.py (:4-21: flags, defaults, default file name, '.pcd' appended when missing)."""
    import argparse
    ap = argparse.ArgumentParser(description="write the 3-face cuboid template as an ASCII PCD v0.7 file")
    ap.add_argument("-L", "--length", default=0.2, type=float, help="cuboid length (m)")
    ap.add_argument("-W", "--width", default=0.1, type=float, help="cuboid width (m)")
    ap.add_argument("-H", "--height", default=0.075, type=float, help="cuboid height (m)")
    ap.add_argument("-d", "--density", default=0.002, type=float, help="sampling step (m)")
    ap.add_argument("-f", "--filename", default="", type=str, help="output filename")
    a = ap.parse_args(argv)
    name = a.filename or template_filename(a.length, a.width, a.height)
    if not name.endswith(".pcd"):
        name += ".pcd"
    data = template_pcd_bytes(a.length, a.width, a.height, a.density)
    with open(name, "wb") as f:
        f.write(data)
    print('Saved "%s" with %d points' % (name, len(make_cuboid_template(a.length, a.width, a.height, a.density))))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
