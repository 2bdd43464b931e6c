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


@pytest.mark.parametrize("method", ["ndt", "vgicp", "gicp"])
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
