"""Frame-by-frame odometry (BASELINE config 4 / SURVEY 8d C4) through the C ABI, following the call pattern of
src/kitti.cpp:95-128: first frame = target; per frame setInputSource -> align -> swapSourceAndTarget ->
pose accumulation.  KITTI is not available offline, so the frames come from the 64-ring LiDAR simulator in
tests/util.py (~118k returns per frame, ApproximateVoxelGrid 0.25 -> ~20k points, kitti.cpp:80-82).

Checked per frame against the fp64 oracle driven through the same sequence (relative pose within 1e-4), and over
the sequence against the simulator's ground truth."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

N_FRAMES = 7


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def frames(O):
    return [O.approx_voxelgrid(util.lidar_frame(i), 0.25) for i in range(N_FRAMES)]


def _gt_rel(i):
    return np.linalg.inv(util.lidar_pose(i - 1)) @ util.lidar_pose(i)


def test_ndt_d2d_odometry_matches_oracle_and_ground_truth(O, frames):
    from fast_gicp_amd import capi
    c = capi.NDTCore(0)
    c.set_distance_mode(1); c.set_neighbor_search_method(1)  # D2D, DIRECT7: NDTCuda defaults (ndt_cuda_impl.hpp:13-16)
    g = O.NDT(mode=1, search=1)
    c.set_target_cloud(frames[0]); g.set_target(frames[0])
    pose, pose_o = np.eye(4), np.eye(4)
    for i in range(1, N_FRAMES):
        c.set_source_cloud(frames[i]); g.set_source(frames[i])
        r, ro = c.align(), g.align()
        assert r["converged"] and ro["converged"]
        assert util.rel_err(r["T"], ro["T"]) < 1e-4, i
        te, re_ = util.pose_error(_gt_rel(i), r["T"])
        assert te < 0.05 and re_ < np.radians(0.5), (i, te, re_)
        c.swap_source_and_target(); g.swap()
        pose, pose_o = pose @ r["T"], pose_o @ ro["T"]
    gt = np.linalg.inv(util.lidar_pose(0)) @ util.lidar_pose(N_FRAMES - 1)
    assert util.rel_err(pose, pose_o) < 1e-4
    te, re_ = util.pose_error(gt, pose)
    assert te < 0.15 and re_ < np.radians(1.0)
    c.close()


def test_vgicp_odometry_matches_oracle(O, frames):
    """Same loop with the FastVGICPCuda core (kitti.cpp:88, commented alternative): device k-NN covariances,
    DIRECT1, resolution 1.0; the swap reuses the source's covariances and rebuilds the voxel map."""
    from fast_gicp_amd import capi
    c = capi.VGICPCore(0)
    c.set_resolution(1.0); c.set_neighbor_search_method(0)
    g = O.FastVGICP(k=20, resolution=1.0, search=0)
    c.set_target_cloud(frames[0]); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
    g.set_target(frames[0])
    for i in range(1, 4):
        c.set_source_cloud(frames[i]); c.find_source_neighbors(20); c.calculate_source_covariances()
        g.set_source(frames[i])
        r, ro = c.align(), g.align()
        assert r["converged"] and ro["converged"]
        assert util.rel_err(r["T"], ro["T"]) < 1e-4, i
        te, re_ = util.pose_error(_gt_rel(i), r["T"])
        assert te < 0.05 and re_ < np.radians(0.5), (i, te, re_)
        c.swap_source_and_target(); g.swap()
    c.close()


@pytest.mark.parametrize("method", ["ndt", "ndt_pipelined", "vgicp", "vgicp_pipelined", "gicp"])
def test_gicp_kitti_app_on_simulated_sequence(tmp_path, method):
    """apps/gicp_kitti (the reference's src/kitti.cpp driver): KITTI-format .bin frames (x, y, z, intensity) in, trajectory
    in KITTI format out; the raw xyzi buffers are downsampled on the device. 5 simulated frames, end pose vs ground truth."""
    import os
    import subprocess
    from fast_gicp_amd import build_host
    exe = build_host.build_kitti()
    n = 5
    for i in range(n):
        f = util.lidar_frame(i)
        np.column_stack([f, np.zeros(len(f), np.float32)]).astype(np.float32).tofile(str(tmp_path / ("%06d.bin" % i)))
    traj = str(tmp_path / "traj.txt")
    out = subprocess.run([exe, str(tmp_path), method, traj], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("fps") == n - 1
    poses = np.loadtxt(traj).reshape(n, 3, 4)
    assert np.allclose(poses[0], np.eye(4)[:3])
    est = np.eye(4); est[:3] = poses[-1]
    gt = np.linalg.inv(util.lidar_pose(0)) @ util.lidar_pose(n - 1)
    te, re_ = util.pose_error(gt, est)
    assert te < 0.12 and re_ < np.radians(1.0), (te, re_)
    if method.endswith("_pipelined"):  # the C++ pipeline (alignAsync / prepareNextSourceDevice / adoptPreparedSource / alignWait of NDTCuda and FastVGICPCuda) == the sequential app
        traj2 = str(tmp_path / "traj_seq.txt")
        out2 = subprocess.run([exe, str(tmp_path), method[:-len("_pipelined")], traj2], capture_output=True, text=True, timeout=300)
        assert out2.returncode == 0, out2.stderr
        assert np.abs(np.loadtxt(traj2) - np.loadtxt(traj)).max() < 1e-6


def _sorted_map(vm):
    coords, num, means, covs = vm
    o = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    return coords[o], num[o], means[o], covs[o]


@pytest.mark.parametrize("mode", [1, 0])  # D2D (NDTCuda default), P2D
def test_pipelined_frame_stream_equals_the_sequential_loop(mode):
    """The two-stage pipeline of the C ABI (fvh_ndt_align_async / _wait, fvh_ndt_prepare_source_device / _adopt_prepared_source, the
    voxel-grid filter on the handle's prepare stream): frame k+1 is filtered, widened and its voxel map built on the second stream while
    the LM kernel of frame k runs (kitti.cpp:95-128: filter -> setInputSource -> align -> swapSourceAndTarget, one stage ahead).
    Same kernels on the same data: the voxel maps it registers are the sequential loop's record for record (fp32 records, compared as
    sets: the ORDER of a map's voxel list is decided by atomics in both loops and differs between two runs of either), iteration counts
    are equal and the poses agree to the noise of that order (fp64 sums in another order: 1e-9 relative is 1e5 x looser than observed)."""
    import torch
    from fast_gicp_amd import capi
    n_frames = 9
    raw = [util.lidar_frame(i) for i in range(n_frames)]
    dev = torch.device("cuda", 0)
    d_raw = [torch.from_numpy(f).to(dev).contiguous() for f in raw]

    def make():
        c = capi.NDTCore(0)
        c.set_distance_mode(mode); c.set_neighbor_search_method(1); c.set_resolution(1.0)
        return c

    # ---- sequential ----
    vg, c = capi.VoxelGrid(0), make()
    ptr, n = vg.filter_device(d_raw[0].data_ptr(), len(raw[0]), 0.25)
    c.set_target_cloud_device(ptr, n, 3)
    seq, seq_maps = [], []
    for i in range(1, n_frames):
        ptr, n = vg.filter_device(d_raw[i].data_ptr(), len(raw[i]), 0.25)
        c.set_source_cloud_device(ptr, n, 3)
        r = c.align()
        assert r["converged"]
        seq.append((r["T"].copy(), r["H"].copy(), r["final_error"], r["num_linearize"], r["num_error_evals"]))
        seq_maps.append((_sorted_map(c.get_voxelmap("source")) if mode == 1 else None, _sorted_map(c.get_voxelmap("target"))))
        c.swap_source_and_target()
    vg.close(); c.close()

    # ---- pipelined ----
    vg, c = capi.VoxelGrid(0), make()
    ptr, n = vg.filter_device(d_raw[0].data_ptr(), len(raw[0]), 0.25)
    c.set_target_cloud_device(ptr, n, 3)
    vg.share_prepare_stream(c)
    ptr, n = vg.filter_device(d_raw[1].data_ptr(), len(raw[1]), 0.25, asynchronous=True)
    c.prepare_source_device(ptr, n, 3)
    for i in range(1, n_frames):
        c.adopt_prepared_source()
        c.align_async()
        if i + 1 < n_frames:  # the next frame, beside the running LM kernel
            ptr, n = vg.filter_device(d_raw[i + 1].data_ptr(), len(raw[i + 1]), 0.25, asynchronous=True)
            c.prepare_source_device(ptr, n, 3)
        r = c.align_wait()
        T, H, err, nl, ne = seq[i - 1]
        assert r["converged"] and r["num_launches"] == 1
        assert (r["num_linearize"], r["num_error_evals"]) == (nl, ne), i
        assert util.rel_err(r["T"], T) < 1e-9 and util.rel_err(r["H"], H) < 1e-9 and abs(r["final_error"] - err) <= 1e-9 * abs(err), i
        src_map, tgt_map = seq_maps[i - 1]
        for a, b in zip(_sorted_map(c.get_voxelmap("target")), tgt_map):
            assert np.array_equal(a, b), i
        if mode == 1:
            for a, b in zip(_sorted_map(c.get_voxelmap("source")), src_map):
                assert np.array_equal(a, b), i
        c.swap_source_and_target()
    # misuse is refused, not undefined
    with pytest.raises(capi.FvhError):
        c.adopt_prepared_source()  # nothing prepared
    with pytest.raises(capi.FvhError):
        c.align_wait()  # nothing in flight
    c.align_async()
    with pytest.raises(capi.FvhError):
        c.align_async()  # one in flight already
    with pytest.raises(capi.FvhError):
        c.swap_source_and_target()
    assert c.align_wait()["converged"] is not None
    vg.close(); c.close()


def _vgicp_voxels(c):
    coords, num, means, covs = c.get_voxelmap()
    o = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    return coords[o], num[o], means[o], covs[o]


@pytest.mark.parametrize("stages,rbf", [(3, False), (2, False), (1, False), (3, True)])
def test_pipelined_vgicp_scan_stream_equals_the_sequential_loop(frames, stages, rbf):
    """fvh_vgicp_prepare_source_device / _adopt_prepared_source / _align_async / _align_wait: scan k+1 is sorted, searched, its covariances
    and (stages = 3) its own voxel map computed on the handle's second stream while the LM kernel of scan k runs; swap_source_and_target()
    then takes the map that came with the scan instead of building one (kitti.cpp:95-128 with FastVGICPCuda, one stage ahead).
    Same kernels on the same data as the sequential calls: equal iteration counts, the voxel map of every registration equal to the
    sequential loop's (counts exactly, means / covariances to the order of their fp64 atomics), poses within 1e-9 (RBF covariances: 1e-7, see below)."""
    import torch
    from fast_gicp_amd import capi
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(np.ascontiguousarray(f, np.float32)).to(dev).contiguous() for f in frames]

    def make():
        c = capi.VGICPCore(0)
        c.set_resolution(1.0); c.set_neighbor_search_method(capi.DIRECT7); c.set_kernel_params(0.5, 2.5)
        return c

    def cov(c, which):
        if rbf:
            getattr(c, "calculate_%s_covariances_rbf" % which)(capi.REG_PLANE)
        else:
            getattr(c, "find_%s_neighbors" % which)(20)
            getattr(c, "calculate_%s_covariances" % which)(capi.REG_PLANE)

    # ---- sequential ----
    c = make()
    c.set_target_cloud_device(d[0].data_ptr(), len(frames[0]), 3); cov(c, "target"); c.create_target_voxelmap()
    seq, seq_maps = [], []
    for i in range(1, N_FRAMES):
        c.set_source_cloud_device(d[i].data_ptr(), len(frames[i]), 3); cov(c, "source")
        r = c.align()
        assert r["converged"]
        seq.append((r["T"].copy(), r["H"].copy(), r["final_error"], r["num_linearize"], r["num_error_evals"]))
        seq_maps.append(_vgicp_voxels(c))
        c.swap_source_and_target()
    c.close()

    # ---- pipelined ----
    c = make()
    c.set_target_cloud_device(d[0].data_ptr(), len(frames[0]), 3); cov(c, "target"); c.create_target_voxelmap()
    c.prepare_source_device(d[1].data_ptr(), len(frames[1]), 3, 20, capi.REG_PLANE, rbf, stages)
    for i in range(1, N_FRAMES):
        c.adopt_prepared_source()
        if stages < 2:
            (c.calculate_source_covariances_rbf if rbf else c.calculate_source_covariances)(capi.REG_PLANE)
        c.align_async()
        if i + 1 < N_FRAMES:  # the next scan, beside the running LM kernel
            c.prepare_source_device(d[i + 1].data_ptr(), len(frames[i + 1]), 3, 20, capi.REG_PLANE, rbf, stages)
        with pytest.raises(capi.FvhError):
            c.get_voxelmap()  # the handle belongs to the running kernel
        r = c.align_wait()
        T, H, err, nl, ne = seq[i - 1]
        assert r["converged"] and r["num_launches"] == 1
        assert (r["num_linearize"], r["num_error_evals"]) == (nl, ne), i
        # (RBF covariances are float sums over the candidates in the cloud's SPATIAL order; a preparation beside a running LM kernel orders the
        #  cloud with the radix passes -- a finer Morton key than the cooperative sort's -- so those sums differ in their last bits: 1e-9 of H observed)
        tol = 1e-7 if rbf else 1e-9
        assert util.rel_err(r["T"], T) < tol and util.rel_err(r["H"], H) < tol and abs(r["final_error"] - err) <= tol * abs(err), i
        got, want = _vgicp_voxels(c), seq_maps[i - 1]
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), i
        assert np.allclose(got[2], want[2], rtol=0, atol=1e-5) and np.allclose(got[3], want[3], rtol=1e-5, atol=1e-7), i
        c.swap_source_and_target()
    assert c.debug_persist_aborts() == 0
    # misuse is refused, not undefined
    with pytest.raises(capi.FvhError):
        c.adopt_prepared_source()  # nothing prepared
    with pytest.raises(capi.FvhError):
        c.align_wait()  # nothing in flight
    with pytest.raises(capi.FvhError):
        c.prepare_source_device(d[0].data_ptr(), len(frames[0]), 3, 20, capi.REG_PLANE, False, 4)
    c.close()


def test_a_voxel_map_is_not_carried_over_a_swap_once_its_cloud_changed(frames):
    """swap_source_and_target() keeps the old target's map with its cloud (it becomes the live map again if the clouds are swapped back)
    -- unless the cloud's covariances changed after the map was built, or the source was replaced: then the reference's rule applies
    (fast_vgicp_cuda.cu:102-104: the map is built from the new target's current covariances)."""
    from fast_gicp_amd import capi
    a, b = frames[0], frames[1]

    def fresh(cloud, reg):
        c = capi.VGICPCore(0)
        c.set_target_cloud(cloud); c.find_target_neighbors(20); c.calculate_target_covariances(reg); c.create_target_voxelmap()
        v = _vgicp_voxels(c)
        c.close()
        return v

    c = capi.VGICPCore(0)
    c.set_target_cloud(a); c.find_target_neighbors(20); c.calculate_target_covariances(capi.REG_PLANE); c.create_target_voxelmap()
    c.set_source_cloud(b); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    c.swap_source_and_target(); c.swap_source_and_target()  # back: the map of `a` was carried
    got, want = _vgicp_voxels(c), fresh(a, capi.REG_PLANE)
    assert np.array_equal(got[0], want[0]) and np.allclose(got[3], want[3], rtol=1e-5, atol=1e-7)
    c.calculate_target_covariances(capi.REG_FROBENIUS)  # the live map stays as built (reference: until create_target_voxelmap) ...
    got = _vgicp_voxels(c)
    assert np.allclose(got[3], want[3], rtol=1e-5, atol=1e-7)
    c.swap_source_and_target(); c.swap_source_and_target()  # ... but across a swap the map follows the cloud's CURRENT covariances
    got, want = _vgicp_voxels(c), fresh(a, capi.REG_FROBENIUS)
    assert np.array_equal(got[0], want[0]) and np.allclose(got[3], want[3], rtol=1e-5, atol=1e-7)
    # a replaced source takes no stale map with it
    c.swap_source_and_target()  # target = b, source = a (map of a carried)
    c.set_source_cloud(frames[2]); c.find_source_neighbors(20); c.calculate_source_covariances(capi.REG_PLANE)
    c.swap_source_and_target()  # target = frames[2]
    got, want = _vgicp_voxels(c), fresh(frames[2], capi.REG_PLANE)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.allclose(got[3], want[3], rtol=1e-5, atol=1e-7)
    c.close()


def test_a_preparation_never_runs_beside_an_lm_grid_that_crowds_the_chip():
    """A persistent LM grid beyond 3/4 of the device's co-resident slots relies on the block -> XCD placement of an otherwise idle chip:
    dispatched beside another stream's kernels it ended in its watchdog (30k points x DIRECT27: 130 aborts of 50 ms in 400 pipelined
    registrations before the rule). The prepared-source calls then queue on the main stream, behind the kernel: no aborts, one launch per
    align, and the poses of the sequential loop."""
    import torch
    from fast_gicp_amd import capi, workloads
    n = 30000
    tgt, src, _ = workloads.synthetic_pair(n, n, seed=7)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(tgt).to(dev).contiguous(), torch.from_numpy(src).to(dev).contiguous()]

    def make():
        c = capi.VGICPCore(0)
        c.set_resolution(0.5); c.set_neighbor_search_method(capi.DIRECT27)
        c.set_target_cloud_device(d[0].data_ptr(), n, 3); c.find_target_neighbors(20); c.calculate_target_covariances(); c.create_target_voxelmap()
        return c

    c = make()
    seq = {}
    c.set_source_cloud_device(d[1].data_ptr(), n, 3); c.find_source_neighbors(20); c.calculate_source_covariances()
    nxt = 0
    for _ in range(2):
        r = c.align()
        seq[1 - nxt] = r["T"].copy()
        c.swap_source_and_target()
        c.set_source_cloud_device(d[nxt].data_ptr(), n, 3); c.find_source_neighbors(20); c.calculate_source_covariances()
        nxt = 1 - nxt
    blocks, cap = c.debug_persist_grid()
    assert blocks * 4 > cap * 3, (blocks, cap)  # (the case this test is about)
    c.close()

    c = make()
    c.prepare_source_device(d[1].data_ptr(), n, 3, 20, capi.REG_PLANE, False, 2)
    c.adopt_prepared_source()
    nxt = 0
    for it in range(60):
        c.align_async()
        c.prepare_source_device(d[nxt].data_ptr(), n, 3, 20, capi.REG_PLANE, False, 2)
        r = c.align_wait()
        assert r["converged"] and r["num_launches"] == 1, it
        assert np.abs(r["T"] - seq[1 - nxt]).max() < 1e-9, it
        c.swap_source_and_target()
        c.adopt_prepared_source()
        nxt = 1 - nxt
    assert c.debug_persist_aborts() == 0
    c.close()
