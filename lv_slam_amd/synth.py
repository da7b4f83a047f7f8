"""Seeded synthetic HDL-64E scan pairs (no KITTI data exists in the build or GPU box).

Sensor model: 64 beams, elevation linear from +2.0 deg to -24.9 deg (the reference hard-codes the
same fan, include/ndt_pca/voxel_grid_covariance_pca.h:97), mounted 1.73 m above a ground plane;
`n_azimuth` firings per revolution (1024 -> 65,536 points, 2048 -> 131,072 points).
Scene: a procedural street along +x (buildings both sides, poles, tree crowns, bushes, parked vehicles), generated per
24 m slot from a hash of (seed, slot) so any frame position sees a deterministic neighbourhood.
Pair k: target = scan at pose P_k = trans(k * 1.0 m, 0, 0); source = scan at P_k * dT_k with
dT_k ~ (x~U(0.6,1.4), y~N(0,0.03), z~N(0,0.01), yaw~N(0,1deg), pitch/roll~N(0,0.2deg)); both scans
are expressed in their own sensor frames, so the true source->target transform is dT_k.
Range noise N(0, 0.02 m).  RNG seed = 0x5EED0000 + pair index.  Points are ordered ring-major
(beam, azimuth).  This is data plumbing for tests/bench, not part of the measured path.
"""
from __future__ import annotations

import functools
import math
import numpy as np
import torch

SENSOR_H = 1.73
SLOT = 24.0
BASE_SEED = 0x5EED0000


@functools.lru_cache(maxsize=4096)
def _slot_primitives(seed: int, slot: int):
    """Primitives of one 24 m street slot (cached: a scan looks at ~11 slots and consecutive scans share all but one; the ~25 scalar
    draws per slot were the largest single CPU cost of generating a benchmark batch).  Callers must not mutate the result."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, slot & 0xFFFFFFFF, 0xC0FFEE])
    x0 = slot * SLOT
    boxes, cyls, sph = [], [], []
    for side in (1.0, -1.0):
        w = rng.uniform(6.0, 20.0)
        depth = rng.uniform(10.0, 15.0)
        h = rng.uniform(4.0, 15.0)
        setback = rng.uniform(8.0, 14.0)
        bx = x0 + rng.uniform(0.0, SLOT - w)
        y_in, y_out = side * setback, side * (setback + depth)
        boxes.append([bx, min(y_in, y_out), 0.0, bx + w, max(y_in, y_out), h])
    for _ in range(4):  # poles / trunks, every second one carries a tree crown
        cx, cy = x0 + rng.uniform(0.0, SLOT), rng.choice([-1.0, 1.0]) * rng.uniform(5.0, 7.5)
        r, hgt = rng.uniform(0.15, 0.4), rng.uniform(3.0, 8.0)
        cyls.append([cx, cy, r, hgt])
        if rng.uniform() < 0.5:
            sph.append([cx, cy, hgt + 0.5, rng.uniform(1.5, 3.0)])
    for _ in range(6):  # bushes / clutter on the verge
        sph.append([x0 + rng.uniform(0.0, SLOT), rng.choice([-1.0, 1.0]) * rng.uniform(4.5, 8.0), rng.uniform(0.2, 0.6),
                    rng.uniform(0.5, 1.4)])
    if rng.uniform() < 0.6:  # parked vehicle
        cx = x0 + rng.uniform(2.5, SLOT - 2.5)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(2.8, 4.5)
        boxes.append([cx - 2.1, cy - 0.9, 0.0, cx + 2.1, cy + 0.9, 1.5])
    return boxes, cyls, sph


def street_primitives(x_center: float, seed: int = BASE_SEED, reach: float = 130.0):
    s0 = int(math.floor((x_center - reach) / SLOT))
    s1 = int(math.floor((x_center + reach) / SLOT))
    boxes, cyls, sph = [], [], []
    for s in range(s0, s1 + 1):
        b, c, p = _slot_primitives(seed, s)
        boxes += b
        cyls += c
        sph += p
    return np.asarray(boxes, dtype=np.float64), np.asarray(cyls, dtype=np.float64), np.asarray(sph, dtype=np.float64)


def beam_directions(n_azimuth: int, device, n_beams: int = 64):
    key = (int(n_azimuth), str(device), int(n_beams))
    d = _BEAMS.get(key)
    if d is None:
        d = _BEAMS[key] = _beam_directions(n_azimuth, device, n_beams)
    return d


_BEAMS: dict = {}


def _beam_directions(n_azimuth: int, device, n_beams: int = 64):
    el = torch.deg2rad(torch.linspace(2.0, -24.9, n_beams, dtype=torch.float64, device=device))
    az = torch.arange(n_azimuth, dtype=torch.float64, device=device) * (2.0 * math.pi / n_azimuth)
    ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
    d = torch.stack([ce * torch.cos(az)[None, :], ce * torch.sin(az)[None, :], se.expand(-1, n_azimuth)], dim=-1)
    return d.reshape(-1, 3)  # ring-major (beam, azimuth)


def rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]], dtype=np.float64)


def cast_scan(pose: np.ndarray, n_azimuth: int, noise: torch.Tensor, ring_u: torch.Tensor, device,
              seed: int = BASE_SEED, n_beams: int = 64) -> torch.Tensor:
    """Ray-cast one scan.  pose = 4x4 sensor->world (world z=0 is the ground; sensor origin at z=1.73
    is folded in here).  Returns float32 [N,3] in the sensor frame."""
    R = torch.as_tensor(pose[:3, :3], dtype=torch.float64, device=device)
    o = torch.as_tensor(pose[:3, 3] + np.array([0, 0, SENSOR_H]), dtype=torch.float64, device=device)
    ds = beam_directions(n_azimuth, device, n_beams)          # sensor-frame unit rays
    d = ds @ R.T                                              # world-frame
    boxes_np, cyls_np, sph_np = street_primitives(float(pose[0, 3]), seed)
    inf = torch.full((d.shape[0],), float("inf"), dtype=torch.float64, device=device)
    # ground z = 0
    t = torch.where(d[:, 2] < -1e-9, -o[2] / d[:, 2], inf)
    # boxes (slab test)
    if len(boxes_np):
        B = torch.as_tensor(boxes_np, device=device)
        inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
        t0 = (B[None, :, 0:3] - o[None, None, :]) * inv[:, None, :]
        t1 = (B[None, :, 3:6] - o[None, None, :]) * inv[:, None, :]
        tn = torch.minimum(t0, t1).amax(dim=-1)
        tf = torch.maximum(t0, t1).amin(dim=-1)
        hit = (tf >= tn) & (tf > 0)
        tb = torch.where(hit, torch.where(tn > 0, tn, tf), inf[:, None].expand(-1, B.shape[0]))
        t = torch.minimum(t, tb.amin(dim=1))
    # vertical cylinders of finite height
    if len(cyls_np):
        C = torch.as_tensor(cyls_np, device=device)
        ox, oy = o[0] - C[None, :, 0], o[1] - C[None, :, 1]
        a = (d[:, 0] ** 2 + d[:, 1] ** 2)[:, None]
        b = 2.0 * (ox * d[:, 0:1] + oy * d[:, 1:2])
        c = ox * ox + oy * oy - C[None, :, 2] ** 2
        disc = b * b - 4 * a * c
        tc = (-b - torch.sqrt(disc.clamp_min(0))) / (2 * a.clamp_min(1e-12))
        zc = o[2] + tc * d[:, 2:3]
        ok = (disc > 0) & (tc > 0) & (zc >= 0) & (zc <= C[None, :, 3])
        tc = torch.where(ok, tc, inf[:, None].expand(-1, C.shape[0]))
        t = torch.minimum(t, tc.amin(dim=1))
    # spheres (tree crowns, bushes)
    if len(sph_np):
        Sp = torch.as_tensor(sph_np, device=device)
        oc = o[None, :] - Sp[:, 0:3]                           # [S,3]
        bq = d @ oc.T                                          # [N,S]  (d . oc)
        cq = (oc * oc).sum(dim=1)[None, :] - Sp[None, :, 3] ** 2
        disc = bq * bq - cq
        ts = -bq - torch.sqrt(disc.clamp_min(0))
        ok = (disc > 0) & (ts > 0)
        ts = torch.where(ok, ts, inf[:, None].expand(-1, Sp.shape[0]))
        t = torch.minimum(t, ts.amin(dim=1))
    rng_ok = (t > 0.5) & (t < 100.0)
    t_noisy = t + noise.to(device)
    pts = ds * torch.where(rng_ok, t_noisy, torch.zeros_like(t))[:, None]
    # fallback: a ground point at horizontal range 4..60 m along the ray's world azimuth
    rr = 4.0 + 56.0 * ring_u.to(device)
    azw = torch.atan2(d[:, 1], d[:, 0])
    pw = torch.stack([o[0] + rr * torch.cos(azw), o[1] + rr * torch.sin(azw), torch.zeros_like(rr)], dim=-1)
    ps = (pw - o[None, :]) @ R                                # R^T (p - o)
    pts = torch.where(rng_ok[:, None], pts, ps)
    return pts.to(torch.float32)


def pair_motion(pair: int, seed: int = BASE_SEED):
    """True source->target transform dT_k (4x4 f64) for pair k."""
    rng = np.random.default_rng([seed & 0xFFFFFFFF, pair & 0xFFFFFFFF, 0xD7])
    fwd = rng.uniform(0.6, 1.4)
    y, z = rng.normal(0, 0.03), rng.normal(0, 0.01)
    yaw, pitch, roll = rng.normal(0, math.radians(1.0)), rng.normal(0, math.radians(0.2)), rng.normal(0, math.radians(0.2))
    T = np.eye(4)
    T[:3, :3] = rot_zyx(yaw, pitch, roll)
    T[:3, 3] = [fwd, y, z]
    return T


def pair_noise(pair: int, n: int, seed: int = BASE_SEED, noise_sigma: float = 0.02, out=None):
    """The seeded draws of make_pair for pair `pair` ([2, n] range noise, [2, n] fallback radii; CPU, f64).  out = (noise, ring): two
    contiguous [2, n] f64 CPU tensors to draw into (bench.py: slices of pinned staging buffers) -- the same draws, in place.
    noise_sigma = None: the unit normals, the caller multiplies (make_pairs(..., unit_noise=True) does, on the device: the generator's
    draws are serial code, the multiplication would be the one OpenMP region of a drawing thread)."""
    g = torch.Generator(device="cpu")
    g.manual_seed((seed + pair) & 0x7FFFFFFFFFFFFFFF)
    if out is None:
        noise = torch.randn(2, n, generator=g, dtype=torch.float64) * noise_sigma
        ring = torch.rand(2, n, generator=g, dtype=torch.float64)
        return noise, ring
    noise, ring = out
    torch.randn(2, n, generator=g, dtype=torch.float64, out=noise)
    if noise_sigma is not None:
        noise.mul_(noise_sigma)
    torch.rand(2, n, generator=g, dtype=torch.float64, out=ring)
    return noise, ring


def make_pair(pair: int, n_azimuth: int = 1024, device="cpu", seed: int = BASE_SEED, n_beams: int = 64,
              noise_sigma: float = 0.02):
    """Returns (target[N,3] f32, source[N,3] f32, dT[4,4] f64 true source->target)."""
    noise, ring = pair_noise(pair, n_azimuth * n_beams, seed, noise_sigma)
    P = np.eye(4)
    P[0, 3] = 1.0 * pair
    dT = pair_motion(pair, seed)
    tgt = cast_scan(P, n_azimuth, noise[0], ring[0], device, seed, n_beams)
    src = cast_scan(P @ dT, n_azimuth, noise[1], ring[1], device, seed, n_beams)
    return tgt, src, dT


def default_guess() -> np.ndarray:
    """Per-pair fixed initial guess: identity with x = +1.0 m, 4x4 f32."""
    G = np.eye(4, dtype=np.float32)
    G[0, 3] = 1.0
    return G


def make_sequence(n_frames: int, n_azimuth: int = 1024, device="cpu", seed: int = BASE_SEED, n_beams: int = 64,
                  noise_sigma: float = 0.02):
    """A drive along the street: frame k is scanned at pose P_k = P_{k-1} * dT_{k-1} (dT from pair_motion).
    Returns (list of [N,3] f32 scans in their own sensor frames, list of 4x4 f64 ground-truth poses P_k)."""
    n = n_azimuth * n_beams
    P = np.eye(4)
    scans, poses = [], []
    for k in range(n_frames):
        g = torch.Generator(device="cpu")
        g.manual_seed((seed + 7919 * (k + 1)) & 0x7FFFFFFFFFFFFFFF)
        noise = torch.randn(n, generator=g, dtype=torch.float64) * noise_sigma
        ring = torch.rand(n, generator=g, dtype=torch.float64)
        scans.append(cast_scan(P, n_azimuth, noise, ring, device, seed, n_beams))
        poses.append(P.copy())
        P = P @ pair_motion(k, seed)
    return scans, poses


# ---- batched generation (bench.py) ------------------------------------------------------------------------------------------------------
# cast_scan issues ~75 small device operations per scan from Python; a benchmark job of 4,541 pairs spent 29 s there, under the GIL, while the GPU
# idled.  cast_scans runs the SAME operations -- same operands, same order, same f64 arithmetic -- over a batch of scans at once: every
# element-wise operation carries a leading batch dimension, the primitive lists are padded to the batch's longest (padding masked out of
# every `where`), and the three K = 3 matrix products stay one call per scan (a batched GEMM may add its three products in another order).
# tests/test_synth.py holds the two bit for bit against each other.

def _within_range(prims, o, reach: float = 101.0):
    """Drop the primitives no ray can meet inside the 100 m the sensor keeps (cast_scan: 0.5 < t < 100): a hit on one of them has
    t >= its distance from the sensor > 100, and cast_scan treats t >= 100 like no hit at all -- the clouds do not change, the
    element-wise passes shrink by a quarter (street_primitives reaches 130-154 m)."""
    boxes, cyls, sph = prims
    if len(boxes):
        gap = np.maximum(np.maximum(boxes[:, 0:3] - o[None, :], o[None, :] - boxes[:, 3:6]), 0.0)
        boxes = boxes[np.sqrt((gap * gap).sum(axis=1)) <= reach]
    if len(cyls):
        cyls = cyls[np.hypot(cyls[:, 0] - o[0], cyls[:, 1] - o[1]) - cyls[:, 2] <= reach]
    if len(sph):
        sph = sph[np.sqrt(((sph[:, 0:3] - o[None, :]) ** 2).sum(axis=1)) - sph[:, 3] <= reach]
    return boxes, cyls, sph


def _pad_prims(lists, width, device):
    """[B][P_b, width] f64 arrays -> ([B, Pmax, width] f64 tensor, [B, Pmax] bool valid)."""
    pmax = max(1, max(len(x) for x in lists))
    buf = np.zeros((len(lists), pmax, width), dtype=np.float64)
    val = np.zeros((len(lists), pmax), dtype=bool)
    for b, x in enumerate(lists):
        if len(x):
            buf[b, :len(x)] = x
            val[b, :len(x)] = True
    return torch.as_tensor(buf, device=device), torch.as_tensor(val, device=device)


def cast_scans(poses, n_azimuth: int, noise: torch.Tensor, ring_u: torch.Tensor, device, seed: int = BASE_SEED, n_beams: int = 64) -> torch.Tensor:
    """cast_scan over a batch: poses = [B] 4x4 arrays, noise / ring_u = [B, N] f64 (any device).  Returns float32 [B, N, 3], every scan
    bit-identical to cast_scan(poses[b], n_azimuth, noise[b], ring_u[b], ...)."""
    nb = len(poses)
    R = torch.as_tensor(np.stack([p[:3, :3] for p in poses]), dtype=torch.float64, device=device)                      # [B,3,3]
    o = torch.as_tensor(np.stack([p[:3, 3] + np.array([0, 0, SENSOR_H]) for p in poses]), dtype=torch.float64, device=device)   # [B,3]
    ds = beam_directions(n_azimuth, device, n_beams)                                                                 # [N,3]
    d = torch.stack([ds @ R[b].T for b in range(nb)])                                                                # [B,N,3]
    prims = [_within_range(street_primitives(float(p[0, 3]), seed), p[:3, 3] + np.array([0, 0, SENSOR_H])) for p in poses]
    n = d.shape[1]
    inf = torch.full((nb, n), float("inf"), dtype=torch.float64, device=device)
    t = torch.where(d[:, :, 2] < -1e-9, -o[:, 2][:, None] / d[:, :, 2], inf)
    if any(len(p[0]) for p in prims):
        B, vB = _pad_prims([p[0] for p in prims], 6, device)                                                         # [B,P,6]
        inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
        t0 = (B[:, None, :, 0:3] - o[:, None, None, :]) * inv[:, :, None, :]
        t1 = (B[:, None, :, 3:6] - o[:, None, None, :]) * inv[:, :, None, :]
        tn = torch.minimum(t0, t1).amax(dim=-1)
        tf = torch.maximum(t0, t1).amin(dim=-1)
        del t0, t1
        hit = (tf >= tn) & (tf > 0) & vB[:, None, :]
        tb = torch.where(hit, torch.where(tn > 0, tn, tf), inf[:, :, None].expand(-1, -1, B.shape[1]))
        t = torch.minimum(t, tb.amin(dim=2))
        del tn, tf, hit, tb
    if any(len(p[1]) for p in prims):
        C, vC = _pad_prims([p[1] for p in prims], 4, device)                                                         # [B,Q,4]
        ox, oy = o[:, 0][:, None, None] - C[:, None, :, 0], o[:, 1][:, None, None] - C[:, None, :, 1]                 # [B,1,Q]
        a = (d[:, :, 0] ** 2 + d[:, :, 1] ** 2)[:, :, None]
        b = 2.0 * (ox * d[:, :, 0:1] + oy * d[:, :, 1:2])
        c = ox * ox + oy * oy - C[:, None, :, 2] ** 2
        disc = b * b - 4 * a * c
        tc = (-b - torch.sqrt(disc.clamp_min(0))) / (2 * a.clamp_min(1e-12))
        zc = o[:, 2][:, None, None] + tc * d[:, :, 2:3]
        ok = (disc > 0) & (tc > 0) & (zc >= 0) & (zc <= C[:, None, :, 3]) & vC[:, None, :]
        tc = torch.where(ok, tc, inf[:, :, None].expand(-1, -1, C.shape[1]))
        t = torch.minimum(t, tc.amin(dim=2))
        del b, disc, tc, zc, ok
    if any(len(p[2]) for p in prims):
        Sp, vS = _pad_prims([p[2] for p in prims], 4, device)                                                        # [B,S,4]
        oc = o[:, None, :] - Sp[:, :, 0:3]                                                                           # [B,S,3]
        bq = torch.zeros(nb, n, Sp.shape[1], dtype=torch.float64, device=device)
        for b in range(nb):
            s = len(prims[b][2])
            if s:
                bq[b, :, :s] = d[b] @ oc[b, :s].T
        cq = (oc * oc).sum(dim=2)[:, None, :] - Sp[:, None, :, 3] ** 2
        disc = bq * bq - cq
        ts = -bq - torch.sqrt(disc.clamp_min(0))
        ok = (disc > 0) & (ts > 0) & vS[:, None, :]
        ts = torch.where(ok, ts, inf[:, :, None].expand(-1, -1, Sp.shape[1]))
        t = torch.minimum(t, ts.amin(dim=2))
        del bq, disc, ts, ok
    rng_ok = (t > 0.5) & (t < 100.0)
    t_noisy = t + noise.to(device)
    pts = ds[None] * torch.where(rng_ok, t_noisy, torch.zeros_like(t))[:, :, None]
    rr = 4.0 + 56.0 * ring_u.to(device)
    azw = torch.atan2(d[:, :, 1], d[:, :, 0])
    pw = torch.stack([o[:, 0][:, None] + rr * torch.cos(azw), o[:, 1][:, None] + rr * torch.sin(azw), torch.zeros_like(rr)], dim=-1)
    pwo = pw - o[:, None, :]
    ps = torch.stack([pwo[b] @ R[b] for b in range(nb)])
    pts = torch.where(rng_ok[:, :, None], pts, ps)
    return pts.to(torch.float32)


def make_pairs(pairs, n_azimuth: int = 1024, device="cpu", seed: int = BASE_SEED, n_beams: int = 64, noise_sigma: float = 0.02, draws=None,
               unit_noise: bool = False):
    """make_pair for a list of pair ids in one batched cast.  Returns (targets [B, N, 3] f32, sources [B, N, 3] f32, [B] dT), bit-identical
    to make_pair(k, ...) for every k.  `draws`: what pair_noise gave for every pair, when the caller drew already (in threads) -- a list
    of (noise, ring) or the two stacked [B, 2, N] tensors (on any device)."""
    n = n_azimuth * n_beams
    if draws is None:
        draws = [pair_noise(k, n, seed, noise_sigma) for k in pairs]
    if isinstance(draws, (list,)):
        noise = torch.stack([x[0] for x in draws])
        ring = torch.stack([x[1] for x in draws])
    else:
        noise, ring = draws
    if unit_noise:                                               # pair_noise(..., noise_sigma=None) drew: the same f64 product, taken here
        noise = noise.to(device) * noise_sigma
    poses, dTs = [], []
    for k in pairs:
        P = np.eye(4)
        P[0, 3] = 1.0 * k
        dT = pair_motion(k, seed)
        poses += [P, P @ dT]
        dTs.append(dT)
    pts = cast_scans(poses, n_azimuth, noise.reshape(2 * len(pairs), n), ring.reshape(2 * len(pairs), n), device, seed, n_beams)
    pts = pts.reshape(len(pairs), 2, n, 3)
    return pts[:, 0], pts[:, 1], dTs
