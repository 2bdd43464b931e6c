"""The spatial (Morton) order every exact search of the engine is culled by, checked DIRECTLY (the k-NN tests only notice a broken order
through wrong neighbour lists -- a valid but unsorted permutation would pass them, slowly): order[] is a permutation; the keys of the sorted
cloud are non-decreasing in the bits the route sorts by, with ties in original-index order (stable LSD passes); the 64-point tile boxes are
the min / max of their points. One case per route: cooperative kernel / its one-workgroup fall-back (n <= 32,768), the two-launch passes
(<= 262,144), the four-launch passes above that."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _spread(v, bits):
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def _keys(pts, axis_bits):
    """kernels_sort.hpp: morton27 (9 bits per axis, large clouds) / the small sorts' 6 bits per axis, on the cloud's bounding cube"""
    lo = pts.min(0)
    extent = np.float32(max((pts.max(0) - lo).max(), np.float32(1e-6)))
    qmax = np.float32((1 << axis_bits) - 1)
    scale = np.float32(np.float32(511.999) / extent) if axis_bits == 9 else np.float32((qmax + np.float32(0.999)) / extent)
    q = np.minimum(qmax, np.maximum(np.float32(0), (pts - lo) * scale)).astype(np.int64)
    return _spread(q[:, 0], axis_bits) | (_spread(q[:, 1], axis_bits) << 1) | (_spread(q[:, 2], axis_bits) << 2)


@pytest.mark.parametrize("n, key_bits, sorted_bits", [
    (1000, 18, 18), (17334, 18, 18), (32768, 18, 18),          # one launch (cooperative) or one workgroup
    (32769, 27, 20), (100000, 27, 20), (262144, 27, 20),       # two launches per pass, top 2 x 10 bits of the 27-bit key
    (262145, 27, 27),                                          # four launches per pass, three 9-bit passes
])
def test_order_is_a_stable_morton_sort_and_tiles_are_boxed(n, key_bits, sorted_bits):
    import os
    from fast_gicp_amd import capi
    if sorted_bits == 20 and os.environ.get("FVH_SORT_FUSED_BITS"):  # (A/B knob of the two-launch passes: 9 -> 18 bits, 0 -> the four-launch passes' 22)
        sorted_bits = {"9": 18, "10": 20, "0": 22}[os.environ["FVH_SORT_FUSED_BITS"]]
    rng = np.random.default_rng(n)
    pts = (rng.normal(size=(n, 3)) * np.array([20.0, 20.0, 2.0])).astype(np.float32)
    pts[: n // 8] = np.round(pts[: n // 8])  # many equal keys and equal points: stability is visible
    c = capi.VGICPCore(0)
    c.set_source_cloud(pts)
    order, boxes = c.debug_spatial_order("source")
    assert np.array_equal(np.sort(order), np.arange(n))
    k = _keys(pts, key_bits // 3)[order] >> (key_bits - sorted_bits)
    d = np.diff(k)
    assert (d >= 0).all(), "keys of the sorted cloud decrease at %d positions" % int((d < 0).sum())
    ties = d == 0
    assert (np.diff(order.astype(np.int64))[ties] > 0).all(), "equal keys are not in original-index order"
    sp = pts[order]
    pad = (-n) % 64
    full = np.concatenate([sp, np.repeat(sp[-1:], pad, 0)]).reshape(-1, 64, 3)
    assert np.array_equal(boxes[:, 0:3], full.min(1)) and np.array_equal(boxes[:, 4:7], full.max(1))
    c.close()


def test_cooperative_sort_is_kept_unless_another_handle_has_a_gang_kernel_in_flight():
    """The one-launch cooperative sort is a gang kernel (32 co-resident workgroups): it is skipped while ANOTHER registration handle of the
    process has a gang kernel IN FLIGHT -- marked at launch under a process-wide lock, in flight until the event recorded behind it has
    fired or that handle's align / synchronize has returned (csrc/host_runtime.inc.hpp: GangRegistry; round 4 guessed with a 20 ms wall-clock window).
    A second handle that merely exists (the reference's align.cpp keeps its NDT object alive while the VGICP rows run) or that the same
    thread uses in turn must not cost the first one its fast sort (it did: +38 us per registration in apps/gicp_align)."""
    from fast_gicp_amd import capi, preprocess
    import os
    tgt, src = preprocess.bundled_pair(os.path.join(util.ROOT, "data"))

    def routes():
        return np.array(capi.debug_sort_routes())

    a, b = capi.VGICPCore(0), capi.VGICPCore(0)
    for c in (a, b):
        c.set_neighbor_search_method(capi.DIRECT7)
    a.synchronize(); b.synchronize()
    r0 = routes()
    # b is alive but idle: a sorts its clouds cooperatively
    a.set_target_cloud(tgt); a.find_target_neighbors(20); a.calculate_target_covariances(); a.create_target_voxelmap()
    a.set_source_cloud(src); a.find_source_neighbors(20); a.calculate_source_covariances()
    ra = a.align()
    assert ra["converged"] and tuple(routes() - r0)[:2] == (2, 0)
    # used in turn by one thread: a's align has returned, so b's sorts are cooperative too
    b.set_target_cloud(tgt); b.find_target_neighbors(20); b.calculate_target_covariances(); b.create_target_voxelmap()
    b.set_source_cloud(src); b.find_source_neighbors(20); b.calculate_source_covariances()
    assert tuple(routes() - r0)[:2] == (4, 0)
    # b's sort is queued and b has not aligned yet: if it is still running when a asks, a takes the radix passes (4 cooperative, 0 one-workgroup,
    # 1 two-launch passes: the one-workgroup sort's 1,024 threads x 120 VGPRs would wait for a CU without LM workgroups, HISTORY.md round 6);
    # if its event has fired already (it lasts ~30 us), a sorts cooperatively (5, 0, 0) -- either way ...
    a.set_source_cloud(src); a.find_source_neighbors(20)
    assert tuple(routes() - r0)[:3] in ((4, 0, 1), (5, 0, 0))
    radix = int((routes() - r0)[2])
    rb = b.align()
    # ... the neighbour lists and the registration do not depend on which spatial order found them
    a.calculate_source_covariances()
    ra2 = a.align()
    assert np.array_equal(ra2["T"], ra["T"]) and np.array_equal(rb["T"], ra["T"])
    # both idle again
    a.set_source_cloud(src); a.find_source_neighbors(20)
    d = routes() - r0
    assert int(d[1]) == 0 and int(d[2]) == radix and int(d[0]) + int(d[2]) == 6
    a.close(); b.close()


def test_sort_mode_1_never_takes_the_cooperative_kernel():
    """FVH_SORT_MODE=1 is what processes SHARING a GPU set (bench.py FVH_BENCH_SHARE_GPU, tests/test_gpu_peer.py, tools/peer_bench.py): the
    in-process GangRegistry cannot see another process's persistent LM grid, so the 32-workgroup cooperative sort must stay off altogether --
    every small sort takes the one-workgroup kernel (the exclusive-cooperative route had swallowed mode 1). The environment is read
    once per process: a child process."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from fast_gicp_amd import capi\n"
        "rng = np.random.default_rng(1)\n"
        "pts = (rng.normal(size=(5000, 3)) * 10).astype(np.float32)\n"
        "c = capi.VGICPCore(0)\n"
        "r0 = np.array(capi.debug_sort_routes())\n"
        "c.set_target_cloud(pts); c.find_target_neighbors(20)\n"
        "c.set_source_cloud(pts[::2].copy()); c.find_source_neighbors(20)\n"
        "c.synchronize()\n"
        "print('ROUTES', *(np.array(capi.debug_sort_routes()) - r0))\n"
    )
    env = dict(os.environ, FVH_SORT_MODE="1", PYTHONPATH=util.ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("ROUTES")][0].split()
    assert (int(line[1]), int(line[2])) == (0, 2), line
