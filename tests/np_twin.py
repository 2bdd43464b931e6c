"""Independent numpy restatement of the oracle's core formulas (test infrastructure only): a second,
differently-written implementation so that the C++ oracle is not its own judge (SURVEY 8c 'oracle plan')."""
import numpy as np


def regularize(C, reg):
    if reg == 0:
        return C
    if reg == 4:
        Ci = np.linalg.inv(C + 1e-3 * np.eye(3))
        return np.linalg.inv(Ci / np.linalg.norm(Ci))
    w, V = np.linalg.eigh(C)
    if reg == 3:
        d = np.array([1e-3, 1.0, 1.0])
    elif reg == 1:
        d = np.maximum(w, 1e-3)
    else:
        d = np.maximum(w / w[2], 1e-3)
    return (V * d) @ V.T


def covariances(xyz, idx, reg):
    p = xyz.astype(np.float64)
    out = np.empty((len(p), 3, 3))
    for i in range(len(p)):
        nb = p[idx[i]]
        d = nb - nb.mean(axis=0)
        out[i] = regularize(d.T @ d / idx.shape[1], reg)
    return out


def voxelmap(xyz, covs, res):
    p = xyz.astype(np.float64)
    keys = np.floor(p / res - 0.5).astype(np.int64)
    vm = {}
    for k, pt, c in zip(map(tuple, keys), p, covs):
        n, m, cc = vm.get(k, (0, np.zeros(3), np.zeros((3, 3))))
        vm[k] = (n + 1, m + pt, cc + c)
    return {k: (n, m / n, c / n) for k, (n, m, c) in vm.items()}


def skew(x):
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])


def linearize(src, src_covs, vm, res, offsets, T):
    R, t = T[:3, :3], T[:3, 3]
    p = src.astype(np.float64)
    q = p @ R.T + t
    coords = np.floor(q / res - 0.5).astype(np.int64)
    H = np.zeros((6, 6)); b = np.zeros(6); err = 0.0; nc = 0
    for i in range(len(p)):
        for o in offsets:
            v = vm.get(tuple(coords[i] + o))
            if v is None:
                continue
            n, mu, Cb = v
            M = np.linalg.inv(Cb + R @ src_covs[i] @ R.T)
            e = mu - q[i]
            w = np.sqrt(n)
            J = np.hstack([skew(q[i]), -np.eye(3)])
            err += w * e @ M @ e
            H += w * J.T @ M @ J
            b += w * J.T @ M @ e
            nc += 1
    return err, H, b, nc
