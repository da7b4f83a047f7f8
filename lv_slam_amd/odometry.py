"""ROS-free restatement of the scan-matching call site of lv_slam's lidar_odometry node.

Mirrors ScanMatchingOdomNodelet::matching_s2k / cloud_callback
(src/lidar_odometry/scan_matching_odom_nodelet.cpp:144-183, 192-261): scan-to-keyframe registration,
constant-velocity guess propagation, keyframe switching, the frame-1 double align, and the KITTI-format
pose row.  It is host-side policy (as it is in the reference) driving any registration object with the
pcl::Registration surface -- normally lv_slam_amd.ndt.NormalDistributionsTransform, i.e. the HIP engine.
"""
from __future__ import annotations

import numpy as np


def quaternionf_w(R: np.ndarray) -> np.float32:
    """w of Eigen::Quaternionf(R.cast<float>()) (Eigen quaternionbase_assign_impl<Matrix3f>), f32 arithmetic."""
    m = np.asarray(R, np.float32)
    t = np.float32(m[0, 0] + m[1, 1] + m[2, 2])
    if t > 0:
        return np.float32(0.5) * np.sqrt(np.float32(t + np.float32(1.0)))
    i = 0
    if m[1, 1] > m[0, 0]:
        i = 1
    if m[2, 2] > m[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(np.float32(m[i, i] - m[j, j] - m[k, k] + np.float32(1.0)))
    return np.float32((m[k, j] - m[j, k]) * (np.float32(0.5) / t))


class ScanMatchingOdometry:
    """matching_s2k state machine.  Parameter defaults are the in-code defaults (:67-76); the KITTI launch file
    uses keyframe_delta_trans = 10 (launch/dlo_kitti.launch:51-53)."""

    def __init__(self, registration, keyframe_delta_trans: float = 5.0, keyframe_delta_angle: float = 0.17,
                 keyframe_delta_time: float = 1.0, tf_velo2cam: np.ndarray | None = None):
        self.reg = registration
        self.keyframe_delta_trans = float(keyframe_delta_trans)
        self.keyframe_delta_angle = float(keyframe_delta_angle)
        self.keyframe_delta_time = float(keyframe_delta_time)
        self.tf_velo2cam = np.eye(4) if tf_velo2cam is None else np.asarray(tf_velo2cam, np.float64)
        self.scan_count = 0
        self.key_id = 0
        self.guess_trans = np.eye(4)
        self.pre_tf_s2k = np.eye(4)
        self.key_pose = np.eye(4)
        self.odom_velo = np.eye(4)
        self.keyframe_stamp = 0.0
        self.n_keyframes = 0
        self.n_aligns = 0

    @staticmethod
    def configure_like_nodelet(reg):
        """registration parameters hard-coded in initialize_params (:109-119): 1.0 m, DIRECT1, eps 0.01, 64 iterations."""
        from . import ndt
        reg.setResolution(1.0)
        reg.setNumThreads(4)
        reg.setNeighborhoodSearchMethod(ndt.DIRECT1)
        reg.setTransformationEpsilon(0.01)
        reg.setMaximumIterations(64)
        return reg

    def matching_s2k(self, stamp: float, cloud) -> np.ndarray:
        """:192-261.  Returns odom_velo (pose of the scan in the frame of the first keyframe), 4x4 f64."""
        if self.scan_count == 0:
            self.reg.setInputTarget(cloud)                         # :197
            self.key_id = 0
            self.guess_trans = np.eye(4)
            self.guess_trans[0, 3] = 1.5                           # :199-200
            self.pre_tf_s2k = np.eye(4)
            self.key_pose = np.eye(4)
            self.keyframe_stamp = stamp
            self.n_keyframes = 1
            return np.eye(4)
        self.reg.setInputSource(cloud)                             # :220
        self.reg.align(self.guess_trans.astype(np.float32))        # :221
        self.n_aligns += 1
        tf_s2k = self.reg.getFinalTransformation().astype(np.float64)
        if self.scan_count == 1:                                   # :223-227: second align seeded with the first result
            self.reg.align(tf_s2k.astype(np.float32))
            self.n_aligns += 1
            tf_s2k = self.reg.getFinalTransformation().astype(np.float64)
        tf_s2s = np.linalg.inv(self.pre_tf_s2k) @ tf_s2k           # :231
        self.odom_velo = self.key_pose @ tf_s2k                    # :234
        dx = float(np.linalg.norm(tf_s2k[:3, 3]))                  # :237
        da = 2.0 * float(np.arccos(np.float64(quaternionf_w(tf_s2k[:3, :3]))))   # :238
        dt = stamp - self.keyframe_stamp
        if dx > self.keyframe_delta_trans or da > self.keyframe_delta_angle or dt > self.keyframe_delta_time:   # :240
            self.reg.setInputTarget(cloud)                         # :243: target rebuild
            self.key_id = self.scan_count
            tf_s2k = np.eye(4)
            self.key_pose = self.odom_velo.copy()
            self.keyframe_stamp = stamp
            self.n_keyframes += 1
        self.pre_tf_s2k = tf_s2k                                   # :249
        self.guess_trans = self.pre_tf_s2k @ tf_s2s                # :250 constant-velocity prior
        return self.odom_velo.copy()

    def cloud_callback(self, stamp: float, cloud):
        """:144-183 without ROS: returns (pose 4x4, KITTI row string)."""
        pose = self.matching_s2k(stamp, cloud)
        odom = self.tf_velo2cam @ pose @ np.linalg.inv(self.tf_velo2cam)     # :156
        row = " ".join("%e" % odom[r, c] for r in range(3) for c in range(4))  # "%le" x 12 (:157-160)
        self.scan_count += 1
        return pose, row


def kitti_row(odom_velo: np.ndarray, tf_velo2cam: np.ndarray | None = None) -> str:
    """One line of the odometry file the nodelet writes (:156-160): odom = tf_velo2cam * pose * tf_velo2cam^-1, twelve "%le" numbers."""
    T = np.eye(4) if tf_velo2cam is None else np.asarray(tf_velo2cam, np.float64)
    odom = T @ np.asarray(odom_velo, np.float64) @ np.linalg.inv(T)
    return " ".join("%e" % odom[r, c] for r in range(3) for c in range(4))


def run_on_device(engine, frames, stamps, keyframe_delta_trans: float = 5.0, keyframe_delta_angle: float = 0.17,
                  keyframe_delta_time: float = 1.0, tf_velo2cam: np.ndarray | None = None):
    """The whole cloud_callback loop (:144-183, 192-261) for a recorded run of frames with the per-frame policy ON THE DEVICE
    (lv_slam_amd.ndt.Engine.sequence_run -> mi355ndt_sequence_run): returns (poses [n,4,4] f64, KITTI rows, per-frame records, stats).
    `engine` carries the registration parameters (configure it like the nodelet: 1.0 m, DIRECT1, eps 0.01, 64 iterations)."""
    recs, stats = engine.sequence_run(frames, stamps, keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time)
    poses = np.stack([r["odom"] for r in recs])
    rows = [kitti_row(p, tf_velo2cam) for p in poses]
    return poses, rows, recs, stats
