"""Deterministic synthetic RealSense D435 frames (the reference ships no recorded data:
cuboid_detection/bags/ is empty, play_rosbag.launch:4 names an absent bag).

Pinhole render at the depth intrinsics of the reference's README.md:74-80 of a table
plane (~0.55 m away, camera pitched ~41 deg, as the rotations in
object_detection/templates/transforms.txt suggest) with 1-3 cuboids of 0.2 x 0.1 x 0.03 m
(iterative_closest_point.launch:39-41) resting on it.  Depth noise sigma = 1 mm (z/0.5)^2,
2 % invalid pixels (NaN xyz), organized 640x480 records of float32 x,y,z + packed rgb
(16 B/point).  Frame i uses seed 20190409 + i.  The random stream is a counter-based
splitmix64 hash, so a frame is a pure function of (seed, pixel).
"""
import numpy as np

WIDTH, HEIGHT = 640, 480
FX = FY = 384.0899353027344
CX, CY = 322.4656982421875, 240.64073181152344
BASE_SEED = 20190409
CUBOID_DIMS = (0.2, 0.1, 0.03)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _uniform(seed, stream, n):
    """n doubles in [0,1), pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(base ^ _splitmix64(idx))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _normal(seed, stream, n):
    u1 = _uniform(seed, stream, n)
    u2 = _uniform(seed, stream + 1, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def _pack_rgb(r, g, b):
    return np.uint32((r << 16) | (g << 8) | b)


def scene_for(index, k_obj=None, seed=None):
    """Scene parameters of frame `index` (a pure function of the seed)."""
    seed = BASE_SEED + index if seed is None else seed
    u = _uniform(seed, 7, 32)
    pitch = np.deg2rad(41.0 + 6.0 * (u[0] - 0.5))
    roll = np.deg2rad(4.0 * (u[1] - 0.5))
    dist = 0.55 + 0.04 * (u[2] - 0.5)
    if k_obj is None:
        k_obj = 1 + int(u[3] * 3.0) % 3
    # table frame in camera coordinates: n = up, e1 ~ camera x, e2 = away from the camera
    n = np.array([np.sin(roll) * np.cos(pitch), -np.cos(roll) * np.cos(pitch), -np.sin(pitch)])
    n /= np.linalg.norm(n)
    e1 = np.array([1.0, 0.0, 0.0]) - n[0] * n
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(e1, n)
    if e2[2] < 0:
        e2 = -e2
    p0 = np.array([0.0, 0.0, dist])
    slots = [0.0, -0.2, 0.2][:k_obj]
    boxes = []
    L, W, H = CUBOID_DIMS
    for k, b0 in enumerate(slots):
        yaw = np.deg2rad(40.0 * (u[8 + 3 * k] - 0.5))
        a = 0.06 * (u[9 + 3 * k] - 0.5)
        b = b0 + 0.02 * (u[10 + 3 * k] - 0.5)
        ex = np.cos(yaw) * e1 + np.sin(yaw) * e2
        ey = np.cross(n, ex)
        c = p0 + a * e1 + b * e2 + (H / 2.0) * n
        R = np.stack([ex, ey, n], axis=1)  # box -> camera
        boxes.append(dict(R=R, c=c, half=np.array([L, W, H]) / 2.0, yaw=yaw))
    return dict(seed=seed, n=n, p0=p0, e1=e1, e2=e2, boxes=boxes)


def scene_grid(index, cols=4, rows=3, dims=(0.05, 0.05, 0.03), pitch=(0.085, 0.085), jitter=0.3):
    """The table of frame `index` with cols x rows small cuboids on it (sizes jittered so that the clusters differ in size):
    more clusters than the CD_MAX_CLUSTERS_PER_FRAME slots of a record, which object_pose_detection.cpp:376 all registers."""
    sc = scene_for(index, k_obj=0)
    u = _uniform(sc["seed"], 11, 4 * cols * rows)
    n, e1, e2, p0 = sc["n"], sc["e1"], sc["e2"], sc["p0"]
    boxes = []
    for r in range(rows):
        for c in range(cols):
            k = r * cols + c
            L, W, H = (d * (1.0 + jitter * (u[4 * k + j] - 0.5)) for j, d in enumerate(dims))
            yaw = np.deg2rad(30.0 * (u[4 * k + 3] - 0.5))
            ex = np.cos(yaw) * e1 + np.sin(yaw) * e2
            ey = np.cross(n, ex)
            ctr = p0 + (c - (cols - 1) / 2.0) * pitch[0] * e1 + (r - (rows - 1) / 2.0) * pitch[1] * e2 + (H / 2.0) * n
            boxes.append(dict(R=np.stack([ex, ey, n], axis=1), c=ctr, half=np.array([L, W, H]) / 2.0, yaw=yaw))
    sc["boxes"] = boxes
    return sc


# BASELINE config 5: five cuboids of distinct dimensions = the five templates: the three dims the reference
# ships templates for + 150x150x50 + 100x100x100 mm; (L, W, H, d = template grid pitch)
CONFIG5_DIMS = ((0.2, 0.1, 0.03, 0.002), (0.2, 0.075, 0.1, 0.005), (0.2, 0.1, 0.075, 0.005), (0.15, 0.15, 0.05, 0.002), (0.1, 0.1, 0.1, 0.002))
CONFIG5_CROP_X = 0.5      # crops widened to the table
CONFIG5_SENSOR = 1000     # 1000 x 1000 virtual sensor = 1 M points per frame


def scene_config5(index):
    """Frame `index` of the config-5 stress: the table with one cuboid of each of CONFIG5_DIMS."""
    sc = scene_for(index, k_obj=0)
    u = _uniform(sc["seed"], 13, 16)
    n, e1, e2, p0 = sc["n"], sc["e1"], sc["e2"], sc["p0"]
    spots = ((-0.26, -0.08), (0.0, -0.10), (0.27, -0.07), (-0.15, 0.17), (0.16, 0.18))
    boxes = []
    for k, ((a, b), (L, W, H, _)) in enumerate(zip(spots, CONFIG5_DIMS)):
        yaw = np.deg2rad(30.0 * (u[3 * k] - 0.5))
        ex = np.cos(yaw) * e1 + np.sin(yaw) * e2
        ey = np.cross(n, ex)
        ctr = p0 + (a + 0.02 * (u[3 * k + 1] - 0.5)) * e1 + (b + 0.02 * (u[3 * k + 2] - 0.5)) * e2 + (H / 2.0) * n
        boxes.append(dict(R=np.stack([ex, ey, n], axis=1), c=ctr, half=np.array([L, W, H]) / 2.0, yaw=yaw))
    sc["boxes"] = boxes
    return sc


def frame_config5(index):
    return render(scene_config5(index), width=CONFIG5_SENSOR, height=CONFIG5_SENSOR)


_RAY_CACHE = {}


def _ray_dirs(width, height):
    key = (width, height)
    if key not in _RAY_CACHE:
        sx, sy = width / float(WIDTH), height / float(HEIGHT)
        v, u = np.divmod(np.arange(width * height), width)
        _RAY_CACHE[key] = ((u - CX * sx) / (FX * sx), (v - CY * sy) / (FY * sy))
    return _RAY_CACHE[key]


def render(scene, width=WIDTH, height=HEIGHT, noise=True, invalid_frac=0.02):
    """(height*width, 4) float32 records x,y,z,rgb(packed uint32 viewed as float32)."""
    npx = width * height
    dx, dy = _ray_dirs(width, height)       # ray direction = (dx, dy, 1)
    nrm, p0 = scene["n"], scene["p0"]
    denom = dx * nrm[0] + dy * nrm[1] + nrm[2]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(denom < -1e-9, (p0 @ nrm) / denom, np.inf)
    color = np.full(npx, _pack_rgb(150, 140, 130), dtype=np.uint32)
    for k, bx in enumerate(scene["boxes"]):
        R = bx["R"]
        o = -(R.T @ bx["c"])                # ray origin (camera centre) in box frame
        tn = np.full(npx, -np.inf)
        tf = np.full(npx, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            for a in range(3):              # slab test, one box axis at a time
                dd = dx * R[0, a] + dy * R[1, a] + R[2, a]
                t1 = (-bx["half"][a] - o[a]) / dd
                t2 = (bx["half"][a] - o[a]) / dd
                np.maximum(tn, np.minimum(t1, t2), out=tn)
                np.minimum(tf, np.maximum(t1, t2), out=tf)
        hit = (tn <= tf) & (tn > 0) & (tn < t)
        t = np.where(hit, tn, t)
        color = np.where(hit, _pack_rgb(200, 30 + 60 * k, 40), color)
    z = t
    seed = scene["seed"]
    if noise:
        z = z + 0.001 * (z / 0.5) ** 2 * _normal(seed, 100, npx)
    bad = ~np.isfinite(z) | (z <= 0)
    if invalid_frac > 0:
        bad |= _uniform(seed, 200, npx) < invalid_frac
    out = np.empty((npx, 4), dtype=np.float32)
    out[:, 0] = dx * z
    out[:, 1] = dy * z
    out[:, 2] = z
    out[bad, :3] = np.nan
    out[:, 3] = color.view(np.float32)
    return out


def frame(index, k_obj=None, width=WIDTH, height=HEIGHT):
    """Frame `index` of the synthetic sequence: (width*height, 4) float32."""
    return render(scene_for(index, k_obj), width, height)


def frames(start, count, k_obj=None, width=WIDTH, height=HEIGHT):
    return np.stack([frame(start + i, k_obj, width, height) for i in range(count)], axis=0)


def truth_poses(scene):
    """4x4 template->camera pose of each cuboid (for the pose-error report)."""
    out = []
    for bx in scene["boxes"]:
        Tm = np.eye(4)
        Tm[:3, :3] = bx["R"]
        Tm[:3, 3] = bx["c"]
        out.append(Tm)
    return out
