"""Synthetic workloads of the benchmark configurations: the LiDAR-like C3/C5 scenes and the C4 stand-in
(a 64-ring spinning-LiDAR simulator; KITTI is not available offline). Pure numpy -- shared by bench.py, the apps and the
tests (tests/util.py re-exports these), so that the product benchmark does not depend on the test tree."""
import functools

import numpy as np


def se3_exp(a):
    """Twist (rot xyz, trans xyz) -> 4x4, rotation first (so3.hpp:80-104 convention); numpy, for workload generation only."""
    a = np.asarray(a, np.float64)
    w, v = a[:3], a[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        R, V = np.eye(3) + W, np.eye(3) + 0.5 * W
    else:
        A, B, Cc = np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3
        R = np.eye(3) + A * W + B * (W @ W)
        V = np.eye(3) + B * W + Cc * (W @ W)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, V @ v
    return T


def pose_error(gt, est):
    """gicp_test.cpp:73-78: (|t|, angle) of gt^-1 * est."""
    d = np.linalg.inv(gt) @ est
    ang = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    return np.linalg.norm(d[:3, 3]), ang


def random_pose(rng, max_angle_deg=2.0, max_trans=0.5):
    w = rng.normal(size=3)
    w *= np.radians(max_angle_deg) * rng.uniform(0.3, 1.0) / np.linalg.norm(w)
    t = rng.normal(size=3)
    t *= max_trans * rng.uniform(0.3, 1.0) / np.linalg.norm(t)
    return se3_exp(np.concatenate([w, t]))


def synthetic_scene(n, seed, extent=60.0, noise=0.02):
    """LiDAR-like scene: ground plane + 12 axis-aligned boxes + 5% clutter, sigma=2 cm."""
    rng = np.random.default_rng(seed)
    boxes_rng = np.random.default_rng(1234)  # the scene itself is fixed; `seed` only drives the sampling
    boxes = []
    for _ in range(12):
        c = boxes_rng.uniform(-extent * 0.8, extent * 0.8, size=2)
        sz = boxes_rng.uniform(3.0, 12.0, size=2)
        hgt = boxes_rng.uniform(2.5, 9.0)
        boxes.append((c, sz, hgt))
    n_clutter = int(0.05 * n)
    n_ground = int(0.55 * n)
    n_walls = n - n_clutter - n_ground
    pts = [np.column_stack([rng.uniform(-extent, extent, n_ground), rng.uniform(-extent, extent, n_ground), np.zeros(n_ground)])]
    per = np.full(12, n_walls // 12)
    per[: n_walls - per.sum()] += 1
    for (c, sz, hgt), m in zip(boxes, per):
        face = rng.integers(0, 4, m)
        u = rng.uniform(-0.5, 0.5, m)
        z = rng.uniform(0, hgt, m)
        x = np.where(face < 2, c[0] + (face * 2 - 1) * sz[0] / 2, c[0] + u * sz[0])
        y = np.where(face < 2, c[1] + u * sz[1], c[1] + ((face - 2) * 2 - 1) * sz[1] / 2)
        pts.append(np.column_stack([x, y, z]))
    pts.append(np.column_stack([rng.uniform(-extent, extent, n_clutter), rng.uniform(-extent, extent, n_clutter), rng.uniform(0, 8, n_clutter)]))
    p = np.concatenate(pts) + rng.normal(scale=noise, size=(n, 3))
    return p[rng.permutation(n)].astype(np.float32)


def synthetic_pair(n_target, n_source, seed=42, extent=60.0):
    """Target + independently sampled source of the same scene moved by T^-1 (ground truth = T)."""
    tgt = synthetic_scene(n_target, seed, extent)
    src_world = synthetic_scene(n_source, seed + 1000, extent)
    T = random_pose(np.random.default_rng(seed + 1))
    Ti = np.linalg.inv(T)
    src = (src_world.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    return tgt, src, T


# ---------------------------------------------------------------------------------------------------
# C4 stand-in: KITTI is not available offline, so a 64-ring spinning-LiDAR simulator over a
# 400 m corridor of the C3-style scene generates the frame sequence src/kitti.cpp:95-128 streams.
# ---------------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=1)
def _corridor_boxes():
    rng = np.random.default_rng(4321)
    boxes = []
    for side in (-1.0, 1.0):
        x = -20.0
        while x < 420.0:
            lx, ly, h = rng.uniform(6.0, 22.0), rng.uniform(5.0, 12.0), rng.uniform(3.0, 12.0)
            off = rng.uniform(7.0, 14.0)
            boxes.append((x, x + lx, side * off if side > 0 else -off - ly, side * off + ly if side > 0 else -off, h))
            x += lx + rng.uniform(1.0, 9.0)
    for _ in range(30):  # parked-car sized clutter near the lane
        x, y = rng.uniform(0, 400), rng.choice([-1, 1]) * rng.uniform(3.5, 6.0)
        boxes.append((x, x + 4.2, y - 0.9, y + 0.9, 1.5))
    return np.array(boxes)


def lidar_pose(i):
    """Ground-truth sensor pose of frame i: 1 m/frame along a gently curving lane, sensor 1.73 m above ground."""
    yaw = 0.15 * np.sin(i / 25.0)
    x, y = float(i), 3.0 * (1 - np.cos(i / 25.0)) * 0.15 * 25.0 / 3.0
    T = np.eye(4)
    T[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
    T[:3, 3] = [x, y, 1.73]
    return T


def lidar_frame(i, seed=7, rings=64, az_steps=1900, max_range=80.0, noise=0.02):
    """Points (float32, sensor frame) of frame i: 64 elevation rings in [-24.8, +2] deg x az_steps azimuths
    (~120k returns), ray-cast against the ground plane and the corridor's boxes, sigma=2 cm range noise."""
    T = lidar_pose(i)
    rng = np.random.default_rng(seed * 100003 + i)
    el = np.radians(np.linspace(-24.8, 2.0, rings))
    az = np.linspace(0, 2 * np.pi, az_steps, endpoint=False) + rng.uniform(0, 2 * np.pi / az_steps)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d_s = np.stack([ce * np.cos(az)[None], ce * np.sin(az)[None], np.broadcast_to(se, (rings, az_steps))], -1).reshape(-1, 3)
    d = d_s @ T[:3, :3].T
    o = T[:3, 3]
    t_hit = np.full(len(d), np.inf)
    dz = d[:, 2]
    tg = np.where(dz < -1e-9, -o[2] / np.where(dz < -1e-9, dz, -1.0), np.inf)
    t_hit = np.minimum(t_hit, tg)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        for x0, x1, y0, y1, h in _corridor_boxes():
            lo = (np.array([x0, y0, 0.0]) - o) * inv
            hi = (np.array([x1, y1, h]) - o) * inv
            tn = np.nanmax(np.minimum(lo, hi), axis=1)
            tf = np.nanmin(np.maximum(lo, hi), axis=1)
            ok = (tn <= tf) & (tn > 0.5)
            t_hit = np.where(ok & (tn < t_hit), tn, t_hit)
    keep = t_hit < max_range
    r = t_hit[keep] + rng.normal(scale=noise, size=int(keep.sum()))
    return (d_s[keep] * r[:, None]).astype(np.float32)
