"""Host-side preprocessing the reference's callers run before the hot path (PCL pieces restated):

  load_pcd                -- pcl::io::loadPCDFile            (src/align.cpp:118-125)
  remove_origin_points    -- the |p|^2 < 1e-3 filter          (src/align.cpp:127-133)
  approximate_voxel_grid  -- pcl::ApproximateVoxelGrid        (src/align.cpp:136-147, src/python/main.cpp:46-62)

PCL is not vendored by the reference and absent here; these follow PCL's published algorithms and
are pinned by README.md:116 (17,249 / 17,518 points on the bundled pair without the origin filter).
Pure numpy, no oracle imports.
"""
import numpy as np


def load_pcd(path):
    """Reads x,y,z (float32) from an ascii/binary .pcd file -> (N,3) float32."""
    with open(path, "rb") as f:
        raw = f.read()
    fields, sizes, types, counts, npoints, mode, pos = [], [], [], [], None, None, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, *vals = line.split()
        if key == "FIELDS":
            fields = vals
        elif key == "SIZE":
            sizes = [int(v) for v in vals]
        elif key == "TYPE":
            types = vals
        elif key == "COUNT":
            counts = [int(v) for v in vals]
        elif key == "POINTS":
            npoints = int(vals[0])
        elif key == "DATA":
            mode = vals[0]
            break
    if not counts:
        counts = [1] * len(fields)
    if mode == "binary":
        dt = []
        for name, s, t, c in zip(fields, sizes, types, counts):
            base = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}[(t, s)]
            dt.append((name, base, (c,)) if c > 1 else (name, base))
        arr = np.frombuffer(raw, dtype=np.dtype(dt), count=npoints, offset=pos)
        return np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float32)
    if mode == "ascii":
        data = np.loadtxt(raw[pos:].decode().splitlines(), dtype=np.float64, ndmin=2)
        cols = np.cumsum([0] + counts)
        ix = [cols[fields.index(a)] for a in "xyz"]
        return data[:npoints, ix].astype(np.float32)
    raise ValueError("unsupported PCD DATA mode: %r" % mode)


def remove_origin_points(xyz):
    """align.cpp:127-133: drop points with squaredNorm() < 1e-3 (fp32 norm)."""
    p = np.asarray(xyz, np.float32)
    sq = p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2]
    return p[sq.astype(np.float64) >= 1e-3]


def approximate_voxel_grid(xyz, leaf):
    """pcl::ApproximateVoxelGrid<PointXYZ> restated: 512-slot history hashed by
    (ix*7171 + iy*3079 + iz*4231) & 511; a slot holding a different voxel is flushed (its fp32
    centroid emitted) before reuse; the remaining slots are flushed in slot order at the end."""
    p = np.ascontiguousarray(xyz, np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(p * inv).astype(np.int64)
    h = ((ijk[:, 0] * 7171 + ijk[:, 1] * 3079 + ijk[:, 2] * 4231) & 511).astype(np.int64)
    ijk_l = ijk.tolist()
    h_l = h.tolist()
    slot_key = [None] * 512
    slot_cnt = [0] * 512
    slot_sum = np.zeros((512, 3), np.float32)
    out = []
    for i in range(len(p)):
        s = h_l[i]
        key = ijk_l[i]
        if slot_cnt[s] and slot_key[s] != key:
            out.append(slot_sum[s] / np.float32(slot_cnt[s]))
            slot_cnt[s] = 0
            slot_sum[s] = 0
        slot_key[s] = key
        slot_cnt[s] += 1
        slot_sum[s] += p[i]
    for s in range(512):
        if slot_cnt[s]:
            out.append(slot_sum[s] / np.float32(slot_cnt[s]))
    return np.asarray(out, np.float32).reshape(-1, 3)


def bundled_pair(data_dir, origin_filter=True, leaf=0.1):
    """The benchmark input of src/align.cpp on data/251370668.pcd (target) / 251371071.pcd (source)."""
    import os
    t = load_pcd(os.path.join(data_dir, "251370668.pcd"))
    s = load_pcd(os.path.join(data_dir, "251371071.pcd"))
    if origin_filter:
        t, s = remove_origin_points(t), remove_origin_points(s)
    return approximate_voxel_grid(t, leaf), approximate_voxel_grid(s, leaf)
