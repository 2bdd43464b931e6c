"""Shared test helpers: bundled-pair preprocessing (via the ORACLE restatement of align.cpp:118-147),
synthetic LiDAR-like scenes, comparison helpers."""
import functools
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


@functools.lru_cache(maxsize=None)
def bundled_pair(origin_filter=True, leaf=0.1, exact_voxelgrid=False):
    """(target, source) float32 Nx3. Default = HEAD align.cpp preprocessing -> 17,047 / 17,334 points."""
    from oracle import oracle as O
    t = O.load_pcd(os.path.join(DATA, "251370668.pcd"))
    s = O.load_pcd(os.path.join(DATA, "251371071.pcd"))
    if origin_filter:
        t, s = O.remove_origin(t), O.remove_origin(s)
    f = O.voxelgrid if exact_voxelgrid else O.approx_voxelgrid
    return f(t, leaf), f(s, leaf)


def relative_pose():
    return np.loadtxt(os.path.join(DATA, "relative.txt"))


def pose_error(gt, est):
    """gicp_test.cpp:73-78: (|t|, angle) of gt^-1 * est."""
    d = np.linalg.inv(gt) @ est
    ang = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    return np.linalg.norm(d[:3, 3]), ang


def random_pose(rng, max_angle_deg=2.0, max_trans=0.5):
    from oracle import oracle as O
    w = rng.normal(size=3)
    w *= np.radians(max_angle_deg) * rng.uniform(0.3, 1.0) / np.linalg.norm(w)
    t = rng.normal(size=3)
    t *= max_trans * rng.uniform(0.3, 1.0) / np.linalg.norm(t)
    return O.se3_exp(np.concatenate([w, t]))


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


def voxel_dict(coords, *arrays):
    return {tuple(c): tuple(a[i] for a in arrays) for i, c in enumerate(np.asarray(coords))}


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
