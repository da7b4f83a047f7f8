"""KITTI odometry velodyne scans as input of the hot path (data plumbing for bench.py / tools; not part of the measured path).

The reference is driven from KITTI odometry sequences (scripts/lidar_odom_kitti.sh:6 plays `<dataset>/sequences/NN`; BASELINE.json
configs 1-4 name seq 04 and seq 00): every frame is `velodyne/%06d.bin`, a flat array of N x 4 little-endian f32 records
(x, y, z, reflectance), N ~ 120 k for the HDL-64E.  Neither this container nor the GPU boxes hold the data set; when a directory is
given (`bench.py --kitti-dir`), consecutive frames (k, k + 1) become the (target, source) pairs, exactly as the odometry node
matches frame k + 1 against frame k before any keyframe logic (scan_matching_odom_nodelet.cpp:197-226).
"""
from __future__ import annotations

import os
import numpy as np


def list_frames(velodyne_dir: str) -> list[str]:
    """Sorted .bin files of a `velodyne/` directory (also accepts the sequence directory that contains it)."""
    d = velodyne_dir
    if os.path.isdir(os.path.join(d, "velodyne")):
        d = os.path.join(d, "velodyne")
    if not os.path.isdir(d):
        raise FileNotFoundError(f"{velodyne_dir}: no such directory")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".bin")]


def load_frame(path: str) -> np.ndarray:
    """One scan as [N, 3] f32 (x, y, z; the reflectance channel is dropped -- NDT never reads it).  A file whose size is no multiple
    of 16 bytes is refused rather than silently truncated."""
    raw = np.fromfile(path, dtype="<f4")
    if raw.size % 4:
        raise ValueError(f"{path}: {raw.size * 4} bytes is not a whole number of x,y,z,reflectance records")
    return np.ascontiguousarray(raw.reshape(-1, 4)[:, :3])


def pack_soa(clouds, device):
    """[(target [n,3], source [m,3]), ...] -> (T, S, target counts, source counts, pitch): two [pair][3][pitch] f32 tensors on `device`
    (the engine's SoA layout, mi355ndt_batch_bind_device), rows zero-padded to the longest cloud rounded up to 64 points."""
    import torch
    tc = [int(len(t)) for t, _ in clouds]
    sc = [int(len(s)) for _, s in clouds]
    pitch = (max(tc + sc + [1]) + 63) // 64 * 64
    T = torch.zeros(len(clouds), 3, pitch, dtype=torch.float32, device=device)
    S = torch.zeros(len(clouds), 3, pitch, dtype=torch.float32, device=device)
    for k, (t, s) in enumerate(clouds):
        if tc[k]:
            T[k, :, :tc[k]] = torch.from_numpy(np.ascontiguousarray(np.asarray(t, np.float32)[:, :3].T)).to(device)
        if sc[k]:
            S[k, :, :sc[k]] = torch.from_numpy(np.ascontiguousarray(np.asarray(s, np.float32)[:, :3].T)).to(device)
    return T, S, tc, sc, pitch


def write_frame(path: str, xyz: np.ndarray, reflectance: float = 0.0) -> None:
    """Inverse of load_frame (tests, and turning synthetic scans into a KITTI-shaped sequence for tools that expect one)."""
    xyz = np.asarray(xyz, np.float32)
    rec = np.empty((len(xyz), 4), "<f4")
    rec[:, :3] = xyz
    rec[:, 3] = reflectance
    rec.tofile(path)
