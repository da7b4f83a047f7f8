"""Call-site policy: CPU test of the ROS-free matching_s2k restatement with a scripted fake
registration (keyframe rule, frame-1 double align, guess propagation, KITTI row), and a GPU test that drives the
HIP engine and an oracle-backed registration through the same policy and compares trajectories."""
import numpy as np
import pytest

from lv_slam_amd.odometry import ScanMatchingOdometry, quaternionf_w


class FakeReg:
    """Pretends every align recovers a fixed forward motion of 1.2 m per frame relative to the current target."""

    def __init__(self):
        self.log = []
        self.target_frame = None
        self.src_frame = None
        self.final = np.eye(4, dtype=np.float32)

    def setInputTarget(self, cloud):
        self.target_frame = int(cloud[0, 0])
        self.log.append(("target", self.target_frame))

    def setInputSource(self, cloud):
        self.src_frame = int(cloud[0, 0])

    def align(self, guess):
        self.log.append(("align", self.src_frame, np.array(guess, np.float64)))
        F = np.eye(4, dtype=np.float32)
        F[0, 3] = 1.2 * (self.src_frame - self.target_frame)
        self.final = F
        return None

    def getFinalTransformation(self):
        return self.final.copy()


def frame(k):
    return np.full((4, 3), float(k), np.float32)


def test_policy_keyframes_guess_and_rows():
    reg = FakeReg()
    od = ScanMatchingOdometry(reg, keyframe_delta_trans=3.0, keyframe_delta_angle=0.17, keyframe_delta_time=1e9)
    poses, rows = [], []
    for k in range(7):
        p, row = od.cloud_callback(0.1 * k, frame(k))
        poses.append(p)
        rows.append(row)
    # frame 0: target only, identity pose
    assert reg.log[0] == ("target", 0) and np.array_equal(poses[0], np.eye(4))
    aligns = [e for e in reg.log if e[0] == "align"]
    # frame 1 is aligned twice; first guess = I with x = 1.5, second seeded with the first result
    assert aligns[0][1] == 1 and aligns[1][1] == 1
    assert aligns[0][2][0, 3] == 1.5 and abs(aligns[1][2][0, 3] - 1.2) < 1e-6
    # constant-velocity guess for frame 2: pre_tf_s2k * tf_s2s = 1.2 + 1.2
    assert abs(aligns[2][2][0, 3] - 2.4) < 1e-6
    # keyframe switch when |t| > 3.0: frame 3 (3.6 m) becomes the new target, guess restarts from tf_s2s
    targets = [e[1] for e in reg.log if e[0] == "target"]
    assert targets == [0, 3, 6]
    assert od.n_keyframes == 3 and od.n_aligns == 7
    # odometry is continuous across the keyframe switch: 1.2 m per frame
    for k in range(7):
        assert abs(poses[k][0, 3] - 1.2 * k) < 1e-5
    # KITTI row: 12 numbers in %e format
    vals = [float(x) for x in rows[4].split()]
    assert len(vals) == 12 and abs(vals[3] - 4.8) < 1e-5 and "e+" in rows[4]


def test_keyframe_time_and_angle_rules():
    reg = FakeReg()
    od = ScanMatchingOdometry(reg, keyframe_delta_trans=1e9, keyframe_delta_angle=1e9, keyframe_delta_time=0.25)
    for k in range(5):
        od.cloud_callback(0.1 * k, frame(k))
    assert [e[1] for e in reg.log if e[0] == "target"] == [0, 3]      # 0.3 s > 0.25 s
    # angle rule uses 2*acos(Quaternionf(R).w())
    c, s = np.cos(0.2), np.sin(0.2)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    assert abs(2 * np.arccos(float(quaternionf_w(R))) - 0.2) < 1e-6
    Rneg = np.diag([1.0, -1.0, -1.0])                                  # trace < 0 branch
    assert abs(float(quaternionf_w(Rneg))) < 1e-6


@pytest.mark.gpu
def test_sequence_gpu_vs_oracle_registration():
    from lv_slam_amd import ndt, synth
    from oracle import oracle_py as O
    from conftest import se3_err

    class OracleReg:
        """the same pcl::Registration surface, backed by the CPU oracle (pclpca, DIRECT1, like the nodelet)"""

        def __init__(self):
            self.prm = O.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=O.DIRECT1,
                                        variant=O.VARIANT_PCA)

        def setInputTarget(self, c):
            self.grid = O.Grid(np.asarray(c, np.float32), self.prm)

        def setInputSource(self, c):
            self.src = np.asarray(c, np.float32)

        def align(self, guess):
            self.r = O.align(self.grid, self.src, np.asarray(guess, np.float32))

        def getFinalTransformation(self):
            return self.r["final"]

    scans, truth = synth.make_sequence(8, 256)
    scans = [s.numpy() for s in scans]
    reg = ScanMatchingOdometry.configure_like_nodelet(ndt.NormalDistributionsTransform(variant=ndt.VARIANT_PCA))
    gpu = ScanMatchingOdometry(reg, keyframe_delta_trans=2.5)
    cpu = ScanMatchingOdometry(OracleReg(), keyframe_delta_trans=2.5)
    for k, s in enumerate(scans):
        pg, rowg = gpu.cloud_callback(0.1 * k, s)
        pc, rowc = cpu.cloud_callback(0.1 * k, s)
        dt, dr = se3_err(pc, pg)
        assert dt < 1e-4 and dr < 1e-5, (k, dt, dr)
    assert gpu.n_keyframes == cpu.n_keyframes and gpu.n_keyframes >= 2 and gpu.n_aligns == cpu.n_aligns == 8
    # and the odometry follows the true drive (scene-noise level over 7 frames)
    dt, dr = se3_err(np.linalg.inv(truth[0]) @ truth[-1], pg)
    assert dt < 0.3 and dr < 0.03, (dt, dr)


def test_kitti_rows_of_a_device_run_match_the_host_policy_rows():
    """lv_slam_amd.odometry.run_on_device formats the device run's poses exactly as cloud_callback does (:156-160); driven here by a fake
    engine that replays the poses of the host-side policy."""
    from lv_slam_amd.odometry import run_on_device, kitti_row
    reg = FakeReg()
    od = ScanMatchingOdometry(reg, keyframe_delta_trans=3.0, keyframe_delta_time=1e9, tf_velo2cam=np.array(
        [[0, -1, 0, 0.1], [0, 0, -1, -0.2], [1, 0, 0, 0.3], [0, 0, 0, 1.0]]))
    poses, rows = [], []
    for k in range(6):
        p, row = od.cloud_callback(0.1 * k, frame(k))
        poses.append(p)
        rows.append(row)

    class FakeEngine:
        def sequence_run(self, frames, stamps, dtr, dan, dti):
            assert (dtr, dan, dti) == (3.0, 0.17, 1e9) and len(frames) == len(stamps) == 6
            return [dict(odom=p) for p in poses], dict(track_ms=1.0)

    P, R, recs, stats = run_on_device(FakeEngine(), [frame(k) for k in range(6)], [0.1 * k for k in range(6)], 3.0, 0.17, 1e9, od.tf_velo2cam)
    assert R == rows and np.array_equal(P, np.stack(poses)) and stats["track_ms"] == 1.0
    assert len(kitti_row(np.eye(4)).split()) == 12
