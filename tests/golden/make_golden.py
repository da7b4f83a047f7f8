#!/usr/bin/env python3
"""Independent NumPy/SciPy restatement of the reference NDT path, used ONLY to pin the C oracle
(oracle/ndt_oracle.c) and to generate the committed fixtures tests/golden/*.npz.

The reference (lv_slam: PCL+Eigen+Sophus) cannot be built or imported here and ships no golden
vectors, so the oracle is pinned against this second, deliberately LITERAL
restatement written straight from the cited lines: it materialises the 4x6 point Jacobian and
the 24x6 point Hessian and evaluates the Eigen expressions as generic left-to-right f32
matrix products (zero terms included), whereas the C oracle uses the algebraically reduced
form.  SE3 exp/log use scipy.linalg.expm/logm and the Newton solve uses numpy SVD, i.e.
different algorithms from the oracle's.  Agreement is therefore evidence against transcription
errors on either side; it is not parity with a reference binary ("parity unpinned").

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
Citations: omp = include/ndt_omp/ndt_omp_impl2.hpp, vgc = include/ndt_omp/voxel_grid_covariance_omp_impl.hpp,
pca = include/ndt_pca/*.
"""
from __future__ import annotations

import os
import sys
import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

f32 = np.float32
f64 = np.float64

KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3


# ----------------------------------------------------------------------------- helpers
def mm32(A, B):
    """Generic f32 matrix product, inner index ascending, every term (also zeros) evaluated."""
    A = np.asarray(A, f32)
    B = np.asarray(B, f32)
    n, k = A.shape
    k2, m = B.shape
    assert k == k2
    out = np.zeros((n, m), f32)
    for i in range(n):
        for j in range(m):
            acc = f32(A[i, 0] * B[0, j])
            for t in range(1, k):
                acc = f32(acc + f32(A[i, t] * B[t, j]))
            out[i, j] = acc
    return out


def gauss_constants(outlier_ratio, resolution):
    """omp:93-100"""
    c1 = 10 * (1 - outlier_ratio)
    c2 = outlier_ratio / float(f32(resolution)) ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2, d3


def hat6(p):
    u, w = p[:3], p[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def se3_exp(p):
    """Sophus::SE3::exp, tangent [upsilon; omega] == matrix exponential of the twist."""
    return scipy.linalg.expm(hat6(np.asarray(p, f64)))


def se3_log(M):
    """Sophus::SE3(R,t).log() == matrix logarithm (R re-orthonormalised as SO3(R) does via the quaternion)."""
    M = np.array(M, f64)
    U, _, Vt = np.linalg.svd(M[:3, :3])
    M[:3, :3] = U @ Vt
    L = np.real(scipy.linalg.logm(M))
    return np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])


# ----------------------------------------------------------------------------- voxel grid
class Leaf:
    __slots__ = ("n", "S", "C", "mean", "cov", "icov", "evals", "label", "weight", "dim2d", "cen", "n_pushed")

    def __init__(self):
        self.n = 0
        self.S = np.zeros(3)          # mean_ accumulator
        self.C = np.eye(3)            # cov_ starts at Identity  (voxel_grid_covariance_omp.h:97-106)
        self.icov = np.zeros((3, 3))
        self.label = 0
        self.weight = 0
        self.dim2d = 0.0
        self.cen = np.zeros(3, f32)   # leaf.centroid, f32 running sum (vgc:229-230, 242-243)
        self.n_pushed = 0


def build_grid(pts, resolution, min_points=6, eig_mult=0.01, pca=False):
    """vgc:48-370 (pca: voxel_grid_covariance_pca_impl.hpp:364-397)."""
    pts = np.asarray(pts, f32)
    leaf = f32(resolution)
    inv = f32(1.0) / leaf
    fin = np.isfinite(pts).all(axis=1)
    mn = pts[fin].min(axis=0)
    mx = pts[fin].max(axis=0)
    min_b = np.floor(mn * inv).astype(np.int64)          # vgc:87-92 (f32 multiply)
    max_b = np.floor(mx * inv).astype(np.int64)
    div_b = max_b - min_b + 1
    mul = np.array([1, div_b[0], div_b[0] * div_b[1]])
    leaves = {}
    for i in range(len(pts)):
        if not fin[i]:
            continue
        p = pts[i]
        ijk = [int(f32(np.floor(f32(p[a] * inv)) - f32(min_b[a]))) for a in range(3)]     # vgc:218-220
        idx = ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2]
        L = leaves.setdefault(int(idx), Leaf())
        p3 = p.astype(f64)
        L.S = L.S + p3                                     # vgc:235
        L.C = L.C + np.outer(p3, p3)                       # vgc:237
        L.cen = (L.cen + p).astype(f32)                    # vgc:242-243
        L.n += 1
    for idx in sorted(leaves):                             # std::map order, vgc:282
        L = leaves[idx]
        L.mean = L.S / L.n                                 # vgc:293
        L.cen = (L.cen / f32(L.n)).astype(f32)             # vgc:289
        L.n_pushed = L.n                                   # what voxel_centroids_ sees (vgc:297-302)
        if L.n >= min_points:
            cov = (L.C - 2 * np.outer(L.S, L.mean)) / L.n + np.outer(L.mean, L.mean)    # vgc:329
            cov = cov * ((L.n - 1.0) / L.n)                                             # vgc:330
            low = np.tril(cov) + np.tril(cov, -1).T        # SelfAdjointEigenSolver reads the lower triangle
            ev, V = np.linalg.eigh(low)                    # ascending
            if ev[0] < 0 or ev[1] < 0 or ev[2] <= 0:       # vgc:337-341
                L.n = -1
                L.cov, L.evals = cov, ev
                continue
            m = eig_mult * ev[2]
            if ev[0] < m:                                  # vgc:345-356 (nested)
                ev = ev.copy()
                ev[0] = m
                if ev[1] < m:
                    ev[1] = m
                cov = V @ np.diag(ev) @ np.linalg.inv(V)
            L.cov, L.evals = cov, ev
            if pca:
                sg = np.sqrt(ev)
                ft = np.array([(sg[2] - sg[1]) / sg[2], (sg[1] - sg[0]) / sg[2], sg[0] / sg[2]])
                L.label = int(np.argmax(ft)) + 1
                scale = {1: 0.75, 2: 1.25, 3: 1.0}[L.label]
                L.dim2d = scale * np.linalg.norm(L.mean)
                L.weight = int(L.dim2d)                    # getDimension2d() returns int (pca.h:222-226)
            L.icov = np.linalg.inv(cov)                    # vgc:359
            if not np.isfinite(L.icov).all():
                L.n = -1
    return dict(leaves=leaves, min_b=min_b, max_b=max_b, div_b=div_b, mul=mul, leaf=leaf, min_points=min_points)


def neighbour_offsets(mode):
    if mode == DIRECT1:
        return [(0, 0, 0)]
    if mode == DIRECT7:                                    # vgc:423-430
        return [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    if mode == DIRECT26:                                   # pcl::getAllNeighborCellIndices (PCL 1.8 voxel_grid.h)
        half = [(i, j, -1) for i in (-1, 0, 1) for j in (-1, 0, 1)] + [(i, -1, 0) for i in (-1, 0, 1)] + [(-1, 0, 0)]
        return half + [(-a, -b, -c) for a, b, c in half]
    raise ValueError(mode)


def radius_search(grid, xt, radius):
    """VoxelGridCovariance::radiusSearch (voxel_grid_covariance_omp.h:505-534): FLANN radius query over the f32
    centroids of the leaves pushed by applyFilter (>= min_points at that time), squared radius float(r*r), strict '<',
    sorted by distance; no nr_points re-check (eigen-failed leaves are returned)."""
    cand = [L for k, L in sorted(grid["leaves"].items()) if L.n_pushed >= grid["min_points"]]
    if not cand:
        return []
    cen = np.array([L.cen for L in cand], f32)
    d = (xt[None, :].astype(f32) - cen).astype(f32)
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(f32) + d[:, 2] * d[:, 2]).astype(f32)
    r2 = f32(float(radius) * float(radius))
    sel = np.nonzero(d2 < r2)[0]
    sel = sel[np.argsort(d2[sel], kind="stable")]
    return [cand[i] for i in sel]


def neighbourhood(grid, xt, mode):
    """vgc:373-404 (DIRECT modes); omp:251-253 (KDTREE)"""
    if mode == KDTREE:
        return radius_search(grid, xt, float(grid["leaf"]))
    leaf = grid["leaf"]
    ijk = [int(np.floor(f32(xt[a] / leaf))) for a in range(3)]     # f32 divide, vgc:379-381
    out = []
    for d in neighbour_offsets(mode):
        c = [ijk[a] + d[a] for a in range(3)]
        if any(c[a] < grid["min_b"][a] or c[a] > grid["max_b"][a] for a in range(3)):
            continue
        idx = sum((c[a] - int(grid["min_b"][a])) * int(grid["mul"][a]) for a in range(3))
        L = grid["leaves"].get(int(idx))
        if L is not None and L.n >= grid["min_points"]:
            out.append(L)
    return out


# ----------------------------------------------------------------------------- sweep
def transform_point(T, p):
    """PCL 1.8 transformPointCloud scalar form."""
    T = np.asarray(T, f32)
    return np.array([f32(f32(f32(f32(T[a, 0] * p[0]) + f32(T[a, 1] * p[1])) + f32(T[a, 2] * p[2])) + T[a, 3])
                     for a in range(3)], f32)


def point_derivatives(x, M32):
    """computePointDerivatives_AngleAxisd, omp:503-532 (f32 overload). M32 = float(exp(p).matrix())."""
    x4 = np.array([[x[0]], [x[1]], [x[2]], [0.0]], f32)
    xt = mm32(M32, x4)[:, 0]
    J = np.zeros((4, 6), f32)
    J[0, 0] = J[1, 1] = J[2, 2] = 1                     # omp:240-241
    J[1, 3] = -xt[2]; J[2, 3] = xt[1]
    J[0, 4] = xt[2]; J[2, 4] = -xt[0]
    J[0, 5] = -xt[1]; J[1, 5] = xt[0]
    Hp = np.zeros((24, 6), f32)
    Hp[12:16, 3] = [0, -xt[1], -xt[2], 0]
    Hp[16:20, 3] = [xt[1], 0, 0, 0]
    Hp[20:24, 3] = [xt[2], 0, 0, 0]
    Hp[12:16, 4] = [0, xt[0], 0, 0]
    Hp[16:20, 4] = [-xt[0], 0, -xt[2], 0]
    Hp[20:24, 4] = [0, xt[2], 0, 0]
    Hp[12:16, 5] = [0, 0, xt[0], 0]
    Hp[16:20, 5] = [0, 0, xt[1], 0]
    Hp[20:24, 5] = [-xt[0], -xt[1], 0, 0]
    return J, Hp


def update_derivatives(g, H, J, Hp, x_trans, c_inv, d1, d2):
    """updateDerivatives, omp:566-619.  g,H are f64 arrays updated in place; returns score_inc (f64)."""
    xt4 = np.array([[f32(x_trans[0]), f32(x_trans[1]), f32(x_trans[2]), f32(0)]], f32)     # 1x4
    c4 = np.zeros((4, 4), f32)
    c4[:3, :3] = c_inv.astype(f32)
    d2f = f32(d2)
    xc = mm32(xt4, c4)                                                # x_trans4 * c_inv4
    dot = mm32(xt4, xc.T)[0, 0]                                       # x_trans4.dot(.)
    arg = f32(f32(f32(-d2f) * dot) * f32(0.5))
    e = f32(np.exp(f64(arg)))                                         # exp in double on the f32 argument -> f32
    score_inc = f32(-d1 * f64(e))
    e = f32(d2f * e)
    if e > 1 or e < 0 or e != e:
        return 0.0
    e = f32(f64(e) * d1)
    cJ = mm32(c4, J)                                                  # 4x6
    v = mm32(xt4, cJ)[0]                                              # 6
    for k in range(6):
        g[k] += f64(f32(e * v[k]))
    JcJ = mm32(J.T, cJ)                                               # 6x6
    for i in range(6):
        z = mm32(xc, Hp[i * 4:(i + 1) * 4, :])[0]
        for j in range(6):
            H[i, j] += f64(f32(e * f32(f32(f32(f32(-d2f) * v[i]) * v[j] + z[j]) + JcJ[j, i])))
    return f64(score_inc)


def sweep(grid, src, T32, M32, d1, d2, mode, pca=False):
    """computeDerivatives, omp:196-305.  T32 transforms the cloud, M32 (4x4 f32) feeds the Jacobian."""
    score = 0.0
    g = np.zeros(6)
    H = np.zeros((6, 6))
    hits = 0
    for p in np.asarray(src, f32):
        if not np.isfinite(p).all():
            continue
        xt = transform_point(T32, p)
        if not np.isfinite(xt).all():        # NaN pose (only reachable through a NaN More-Thuente trial value): the reference's
            continue                         # float->int cast is UB there; canonical choice = such a point has no neighbours
        s_pt, g_pt, H_pt = 0.0, np.zeros(6), np.zeros((6, 6))
        for L in neighbourhood(grid, xt, mode):
            x_trans = xt.astype(f64) - L.mean                          # omp:276-279
            J, Hp = point_derivatives(p, M32)
            s_pt += update_derivatives(g_pt, H_pt, J, Hp, x_trans, L.icov, d1, d2)
            if pca:                                                    # ndt_pca_impl2.hpp:295-296
                w = float(L.weight)
                s_pt *= w; g_pt *= w; H_pt *= w
            hits += 1
        score += s_pt; g += g_pt; H += H_pt
    return score, g, H, hits


def cmin(a, b):
    """std::min(a, b) = (b < a) ? b : a  -- a NaN first argument is returned as is"""
    return b if b < a else a


def cmax(a, b):
    """std::max(a, b) = (a < b) ? b : a"""
    return b if a < b else a


def update_interval_mt(I, a_t, f_t, g_t):
    """updateIntervalMT, omp:717-755.  I = [a_l, f_l, g_l, a_u, f_u, g_u] updated in place; returns interval_converged."""
    if f_t > I[1]:
        I[3], I[4], I[5] = a_t, f_t, g_t
        return False
    if g_t * (I[0] - a_t) > 0:
        I[0], I[1], I[2] = a_t, f_t, g_t
        return False
    if g_t * (I[0] - a_t) < 0:
        I[3], I[4], I[5] = I[0], I[1], I[2]
        I[0], I[1], I[2] = a_t, f_t, g_t
        return False
    return True


def trial_value_selection_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t):
    """trialValueSelectionMT, omp:758-838 (IEEE semantics: sqrt of a negative and 0/0 give NaN, never an exception)."""
    with np.errstate(all="ignore"):
        a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t = (f64(v) for v in (a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t))
        if f_t > f_l:
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t))
            return float(a_c) if abs(a_c - a_l) < abs(a_q - a_l) else float(0.5 * (a_q + a_c))
        if g_t * g_l < 0:
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
            return float(a_c) if abs(a_c - a_t) >= abs(a_s - a_t) else float(a_s)
        if abs(g_t) <= abs(g_l):
            z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l
            w = np.sqrt(z * z - g_t * g_l)
            a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w)
            a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l
            a_n = a_c if abs(a_c - a_t) < abs(a_s - a_t) else a_s
            if a_t > a_l:
                return float(cmin(a_t + 0.66 * (a_u - a_t), a_n))
            return float(cmax(a_t + 0.66 * (a_u - a_t), a_n))
        z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u
        w = np.sqrt(z * z - g_t * g_u)
        return float(a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w))


def compute_hessian(grid, src, T32, xt_tangent, d1, d2):
    """computeHessian + updateHessian, omp:622-714: f64 throughout, kd-tree neighbourhoods whatever the search method,
    point derivatives from the f64 overload omp:535-563 (x_t = exp(p) * (x, 0) in double)."""
    H = np.zeros((6, 6))
    R = se3_exp(xt_tangent)[:3, :3]
    for p in np.asarray(src, f32):
        if not np.isfinite(p).all():
            continue
        xt = transform_point(T32, p)
        if not np.isfinite(xt).all():
            continue
        x = p.astype(f64)
        r = R @ x
        J = np.zeros((3, 6)); J[:, :3] = np.eye(3)
        J[1, 3] = -r[2]; J[2, 3] = r[1]; J[0, 4] = r[2]; J[2, 4] = -r[0]; J[0, 5] = -r[1]; J[1, 5] = r[0]
        Hp = np.zeros((18, 6))
        Hp[9:12, 3] = [0, -r[1], -r[2]]; Hp[12:15, 3] = [r[1], 0, 0]; Hp[15:18, 3] = [r[2], 0, 0]
        Hp[9:12, 4] = [0, r[0], 0]; Hp[12:15, 4] = [-r[0], 0, -r[2]]; Hp[15:18, 4] = [0, r[2], 0]
        Hp[9:12, 5] = [0, 0, r[0]]; Hp[12:15, 5] = [0, 0, r[1]]; Hp[15:18, 5] = [-r[0], -r[1], 0]
        for L in radius_search(grid, xt, float(grid["leaf"])):
            x_trans = xt.astype(f64) - L.mean
            C = L.icov
            with np.errstate(all="ignore"):
                e = d2 * np.exp(-d2 * (x_trans @ (C @ x_trans)) / 2)
            if e > 1 or e < 0 or e != e:
                continue
            e *= d1
            for i in range(6):
                cov_dxd = C @ J[:, i]
                for j in range(6):
                    H[i, j] += e * (-d2 * (x_trans @ cov_dxd) * (x_trans @ (C @ J[:, j])) +
                                    x_trans @ (C @ Hp[3 * i:3 * i + 3, j]) + J[:, j] @ cov_dxd)
    return H


def step_length_mt(grid, src, x, dirn, step_init, step_max, step_min, score, g, H, prm, d1, d2):
    """computeStepLengthMT omp:841-1003, literal (every trial re-sweeps).  Returns (a_t, dirn, score, g, H, final32, hits, info);
    final32 is None when the function returned before touching final_transformation_ (omp:856-857)."""
    mode, pca = prm["neighbor_mode"], prm["variant"] == 1
    phi_0 = -score
    d_phi_0 = -(g @ dirn)
    if d_phi_0 >= 0:
        if d_phi_0 == 0:
            return 0.0, dirn, score, g, H, None, None, dict(step_iterations=0)
        d_phi_0 *= -1
        dirn = -dirn
    mu, nu = 1.e-4, 0.9
    psi = lambda a, f_a: f_a - phi_0 - mu * d_phi_0 * a              # auxilaryFunction_PsiMT  ndt_omp.h:480-483
    dpsi = lambda g_a: g_a - mu * d_phi_0                            # auxilaryFunction_dPsiMT ndt_omp.h:493-496
    I = [0.0, psi(0.0, phi_0), dpsi(d_phi_0), 0.0, psi(0.0, phi_0), dpsi(d_phi_0)]
    interval_converged = (step_max - step_min) > 0
    open_interval = True
    a_t = cmax(cmin(step_init, step_max), step_min)
    with np.errstate(all="ignore"):
        x_t = x + dirn * a_t
        final = se3_exp(x_t).astype(f32)
    score, g, H, hits = sweep(grid, src, final, final, d1, d2, mode, pca)
    phi_t, d_phi_t = -score, -(g @ dirn)
    psi_t, d_psi_t = psi(a_t, phi_t), dpsi(d_phi_t)
    it = 0
    while not interval_converged and it < 10 and not (psi_t <= 0 and d_phi_t <= -nu * d_phi_0):
        if open_interval:
            a_t = trial_value_selection_mt(*I, a_t, psi_t, d_psi_t)
        else:
            a_t = trial_value_selection_mt(*I, a_t, phi_t, d_phi_t)
        a_t = cmax(cmin(a_t, step_max), step_min)
        with np.errstate(all="ignore"):
            x_t = x + dirn * a_t
            final = se3_exp(x_t).astype(f32) if np.isfinite(x_t).all() else np.full((4, 4), np.nan, f32)
        score, g, _, hits = sweep(grid, src, final, final, d1, d2, mode, pca)     # compute_hessian = false: H comes back zero
        H = np.zeros((6, 6))
        phi_t, d_phi_t = -score, -(g @ dirn)
        psi_t, d_psi_t = psi(a_t, phi_t), dpsi(d_phi_t)
        if open_interval and (psi_t <= 0 and d_psi_t >= 0):
            open_interval = False
            I[1] = I[1] + phi_0 - mu * d_phi_0 * I[0]; I[2] = I[2] + mu * d_phi_0
            I[4] = I[4] + phi_0 - mu * d_phi_0 * I[3]; I[5] = I[5] + mu * d_phi_0
        if open_interval:
            interval_converged = update_interval_mt(I, a_t, psi_t, d_psi_t)
        else:
            interval_converged = update_interval_mt(I, a_t, phi_t, d_phi_t)
        it += 1
    if it:
        H = compute_hessian(grid, src, final, x_t, d1, d2)
    return a_t, dirn, score, g, H, final, hits, dict(step_iterations=it)


def align(grid, src, guess32, prm):
    """computeTransformation omp:87-188 + computeStepLengthMT omp:841-1003 (its loop is live iff step_size <= eps/2)."""
    if not (prm["step_size"] - prm["trans_epsilon"] / 2 > 0):
        return align_mt_live(grid, src, guess32, prm)
    return align_dead_mt(grid, src, guess32, prm)


def align_mt_live(grid, src, guess32, prm):
    d1, d2, _ = gauss_constants(prm["outlier_ratio"], prm["resolution"])
    eps, step = prm["trans_epsilon"], prm["step_size"]
    mode, pca = prm["neighbor_mode"], prm["variant"] == 1
    final = np.array(guess32, f32)
    p = se3_log(final.astype(f64))
    score, g, H, hits = sweep(grid, src, final, se3_exp(p).astype(f32), d1, d2, mode, pca)
    it, trace, mt_its = 0, [(score, g.copy(), H.copy())], []
    while True:
        U, S, Vt = np.linalg.svd(H)
        rank = int((S >= max(S[0] * 6 * np.finfo(f64).eps, np.finfo(f64).tiny)).sum()) if S[0] > 0 else 0
        dp = Vt[:rank].T @ ((U[:, :rank].T @ (-g)) / S[:rank])
        n = np.linalg.norm(dp)
        if n == 0 or n != n:
            return dict(final=final, iterations=it, converged=bool(n == n), score=score, trace=trace, hits=hits, mt_its=mt_its)
        dp = dp / n
        a, dp, score, g, H, f2, h2, info = step_length_mt(grid, src, p, dp, n, step, eps / 2, score, g, H, prm, d1, d2)
        if f2 is not None:
            final, hits = f2, h2
            trace.append((score, g.copy(), H.copy()))
        mt_its.append(info["step_iterations"])
        with np.errstate(all="ignore"):
            dpv = dp * a
            p = se3_log(se3_exp(dpv) @ se3_exp(p)) if np.isfinite(dpv).all() else np.full(6, np.nan)
        conv = it > prm["max_iterations"] or (it and abs(a) < eps)
        it += 1
        if conv:
            return dict(final=final, iterations=it, converged=True, score=score, trace=trace, hits=hits, mt_its=mt_its)


def align_dead_mt(grid, src, guess32, prm):
    """computeTransformation omp:87-188 + the live prefix of computeStepLengthMT omp:841-907."""
    d1, d2, _ = gauss_constants(prm["outlier_ratio"], prm["resolution"])
    eps, step = prm["trans_epsilon"], prm["step_size"]
    mode, pca = prm["neighbor_mode"], prm["variant"] == 1
    final = np.array(guess32, f32)
    p = se3_log(final.astype(f64))
    score, g, H, hits = sweep(grid, src, final, se3_exp(p).astype(f32), d1, d2, mode, pca)
    it, trace = 0, [(score, g.copy(), H.copy())]
    while True:
        U, S, Vt = np.linalg.svd(H)                                    # JacobiSVD.solve: thresholded pinv
        rank = int((S >= max(S[0] * 6 * np.finfo(f64).eps, np.finfo(f64).tiny)).sum()) if S[0] > 0 else 0
        dp = Vt[:rank].T @ ((U[:, :rank].T @ (-g)) / S[:rank])
        n = np.linalg.norm(dp)
        if n == 0 or n != n:
            return dict(final=final, iterations=it, converged=bool(n == n), score=score, trace=trace, hits=hits)
        dp = dp / n
        dphi0 = -(g @ dp)
        if dphi0 >= 0 and dphi0 == 0:
            a = 0.0
        else:
            if dphi0 >= 0:
                dp = -dp
            a = max(min(n, step), eps / 2)
            xt = p + dp * a
            final = se3_exp(xt).astype(f32)
            score, g, H, hits = sweep(grid, src, final, final, d1, d2, mode, pca)
            trace.append((score, g.copy(), H.copy()))
        dpv = dp * a
        p = se3_log(se3_exp(dpv) @ se3_exp(p))
        conv = it > prm["max_iterations"] or (it and abs(a) < eps)
        it += 1
        if conv:
            return dict(final=final, iterations=it, converged=True, score=score, trace=trace, hits=hits)


# ----------------------------------------------------------------------------- fixtures
def leaves_to_arrays(grid):
    keys = sorted(grid["leaves"])
    L = [grid["leaves"][k] for k in keys]
    valid = [i for i, l in enumerate(L) if l.n >= grid["min_points"]]
    return dict(
        leaf_idx=np.array(keys, np.int64), leaf_n=np.array([l.n for l in L], np.int64),
        leaf_mean=np.array([l.mean for l in L]),
        v_sel=np.array(valid, np.int64),
        v_cov=np.array([L[i].cov for i in valid]), v_icov=np.array([L[i].icov for i in valid]),
        v_evals=np.array([L[i].evals for i in valid]),
        v_label=np.array([L[i].label for i in valid], np.int64), v_weight=np.array([L[i].weight for i in valid], np.int64),
        min_b=np.asarray(grid["min_b"], np.int64), max_b=np.asarray(grid["max_b"], np.int64), div_b=np.asarray(grid["div_b"], np.int64))


def make_case(name, pair, n_az, n_beams, resolution, mode, variant, n_src_sweep=200, n_src_align=500):
    from lv_slam_amd import synth
    tgt, src, dT = synth.make_pair(pair, n_az, n_beams=n_beams)
    tgt, src = tgt.numpy(), src.numpy()
    prm = dict(resolution=resolution, step_size=0.1, outlier_ratio=0.55, trans_epsilon=0.01, max_iterations=64,
               neighbor_mode=mode, variant=variant, min_points_per_voxel=6, min_covar_eigvalue_mult=0.01)
    grid = build_grid(tgt, resolution, pca=(variant == 1))
    out = leaves_to_arrays(grid)
    d1, d2, d3 = gauss_constants(0.55, resolution)
    out.update(gauss=np.array([d1, d2, d3]))
    # deterministic source subsets (strided so they cover the scene)
    s_sweep = src[:: max(1, len(src) // n_src_sweep)][:n_src_sweep]
    s_align = src[:: max(1, len(src) // n_src_align)][:n_src_align]
    guess = synth.default_guess()
    p0 = se3_log(guess.astype(f64))
    p1 = p0 + np.array([0.03, -0.02, 0.01, 0.004, -0.003, 0.01])
    sweeps = []
    for p in (p0, p1):
        M32 = se3_exp(p).astype(f32)
        sc, g, H, hits = sweep(grid, s_sweep, M32, M32, d1, d2, mode, variant == 1)
        sweeps.append((p, M32, sc, g, H, hits))
    out.update(sweep_p=np.array([s[0] for s in sweeps]), sweep_T=np.array([s[1] for s in sweeps]),
               sweep_score=np.array([s[2] for s in sweeps]), sweep_g=np.array([s[3] for s in sweeps]),
               sweep_H=np.array([s[4] for s in sweeps]), sweep_hits=np.array([s[5] for s in sweeps], np.int64))
    r = align(grid, s_align, guess, prm)
    out.update(align_final=r["final"], align_iterations=np.int64(r["iterations"]), align_converged=np.int64(r["converged"]),
               align_score=np.float64(r["score"]), align_hits=np.int64(r["hits"]),
               align_trace_score=np.array([t[0] for t in r["trace"]]),
               align_trace_g=np.array([t[1] for t in r["trace"]]), align_trace_H=np.array([t[2] for t in r["trace"]]))
    out.update(target=tgt, src_sweep=s_sweep, src_align=s_align, guess=guess, true_dT=dT,
               params=np.array([resolution, 0.1, 0.55, 0.01, 64, mode, variant, 6, 0.01]))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: target {len(tgt)} pts, leaves {len(out['leaf_idx'])}, valid {len(out['v_sel'])}, "
          f"sweep hits {out['sweep_hits']}, align it {r['iterations']} score {r['score']:.6f} -> {os.path.getsize(path)} B")


def se3_vectors():
    rng = np.random.default_rng(7)
    ps = np.concatenate([rng.normal(0, 1, (24, 6)) * np.array([2, 2, 2, 0.5, 0.5, 0.5]),
                         rng.normal(0, 1, (8, 6)) * np.array([1, 1, 1, 1e-6, 1e-6, 1e-6]),
                         np.zeros((1, 6))])
    Ms = np.array([se3_exp(p) for p in ps])
    dps = rng.normal(0, 0.05, (len(ps), 6))
    comp = np.array([se3_log(se3_exp(d) @ se3_exp(p)) for d, p in zip(dps, ps)])
    np.savez_compressed(os.path.join(HERE, "se3_vectors.npz"), p=ps, M=Ms, dp=dps, compose_log=comp)
    print("se3_vectors:", len(ps))


def make_mt_case(name, pair, n_az, n_beams, mode, variant, n_src=400):
    """step_size <= eps/2: the More-Thuente loop and computeHessian are live (omp:888, 920-1000)."""
    from lv_slam_amd import synth
    tgt, src, dT = synth.make_pair(pair, n_az, n_beams=n_beams)
    tgt, src = tgt.numpy(), src.numpy()
    base = dict(resolution=1.0, step_size=0.1, outlier_ratio=0.55, trans_epsilon=0.01, max_iterations=64,
                neighbor_mode=mode, variant=variant, min_points_per_voxel=6, min_covar_eigvalue_mult=0.01)
    live = dict(base, step_size=0.004)
    grid = build_grid(tgt, 1.0, pca=(variant == 1))
    d1, d2, _ = gauss_constants(0.55, 1.0)
    s_al = src[:: max(1, len(src) // n_src)][:n_src]
    g_far = synth.default_guess()
    r0 = align(grid, s_al, g_far, base)                      # an ordinary registration: its result is the "near" guess
    g_near = r0["final"]
    g_over = g_near.copy()
    g_over[0, 3] -= f32(0.002)                               # a hair off the optimum: the forced step eps/2 overshoots
    out = dict(target=tgt, src_align=s_al, guess_far=g_far, guess_near=g_near, guess_over=g_over, params=np.array([1.0, 0.004, 0.55, 0.01, 64, mode, variant, 6, 0.01]))
    for tag, G in (("far", g_far), ("near", g_near), ("over", g_over)):
        r = align(grid, s_al, G, live)
        out.update({f"{tag}_final": r["final"], f"{tag}_iterations": np.int64(r["iterations"]), f"{tag}_converged": np.int64(r["converged"]),
                    f"{tag}_score": np.float64(r["score"]), f"{tag}_mt_its": np.array(r["mt_its"], np.int64),
                    f"{tag}_trace_H": np.array([t[2] for t in r["trace"]]), f"{tag}_trace_g": np.array([t[1] for t in r["trace"]])})
        print(f"{name}/{tag}: it {r['iterations']} conv {r['converged']} MT loop iterations {r['mt_its']} score {r['score']:.6f}")
    p1 = se3_log(g_far.astype(f64)) + np.array([0.03, -0.02, 0.01, 0.004, -0.003, 0.01])
    T1 = se3_exp(p1).astype(f32)
    out.update(hess_p=p1, hess_H=compute_hessian(grid, s_al, T1, p1, d1, d2))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: -> {os.path.getsize(path)} B")


def calculate_score(grid, cloud, d1, d2, d3):
    """omp:1006-1040 (pca: ndt_pca_impl2.hpp:1013-1047), literally: `cloud` is the already transformed cloud; f64 throughout."""
    score = 0.0
    for p in np.asarray(cloud, f32):
        hood = radius_search(grid, p, float(grid["leaf"]))              # omp:1018
        for L in hood:
            x_trans = p.astype(f64) - L.mean                            # omp:1025-1028
            e_x_cov_x = np.exp(-d2 * x_trans.dot(L.icov @ x_trans) / 2)  # omp:1033
            score_inc = -d1 * e_x_cov_x - d3                            # omp:1035
            score += score_inc / len(hood)                              # omp:1037
    return score / len(cloud)                                           # omp:1040


def convert_transform(x):
    """ndt_omp.h:209-228 through an independent route: scipy rotations in f64, cast at the end (Eigen works in f32 throughout)."""
    from scipy.spatial.transform import Rotation as Rot
    M = np.eye(4)
    M[:3, :3] = (Rot.from_euler("x", float(f32(x[3]))) * Rot.from_euler("y", float(f32(x[4]))) * Rot.from_euler("z", float(f32(x[5])))).as_matrix()
    M[:3, 3] = [float(f32(v)) for v in x[:3]]
    return M.astype(f32)


def make_score_cases():
    """calculateScore / convertTransform vectors on the clouds of two committed fixtures (nothing else is rewritten)."""
    out = {}
    for tag, name, variant in (("omp", "omp_direct7_r1", 0), ("pca", "pca_direct1_r05", 1)):
        fx = np.load(os.path.join(HERE, name + ".npz"))
        res = float(fx["params"][0])
        grid = build_grid(fx["target"], res, pca=(variant == 1))
        T = fx["align_final"].astype(f32)
        cloud = np.array([transform_point(T, p) for p in fx["src_align"]], f32)
        ctor = gauss_constants(0.55, 1.0)                                # the constructor's members (omp:70-76)
        own = gauss_constants(0.55, res)                                 # after an align() with the fixture's parameters (omp:93-100)
        out.update({f"{tag}_fixture": np.array(name), f"{tag}_cloud": cloud, f"{tag}_gauss_ctor": np.array(ctor), f"{tag}_gauss_align": np.array(own),
                    f"{tag}_score_ctor": np.float64(calculate_score(grid, cloud, *ctor)), f"{tag}_score_align": np.float64(calculate_score(grid, cloud, *own)),
                    f"{tag}_score_raw_source": np.float64(calculate_score(grid, fx["src_align"], *own))})
        print(tag, out[f"{tag}_score_ctor"], out[f"{tag}_score_align"], out[f"{tag}_score_raw_source"])
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.normal(0, 1, (24, 6)) * np.array([5, 5, 5, 1.5, 1.5, 1.5]), np.zeros((1, 6)), [[1, 2, 3, np.pi, -np.pi / 2, 0.25]]])
    out.update(ct_x=xs, ct_M=np.array([convert_transform(x) for x in xs]))
    np.savez_compressed(os.path.join(HERE, "calc_score.npz"), **out)
    print("calc_score.npz:", os.path.getsize(os.path.join(HERE, "calc_score.npz")), "B")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "score":       # calculateScore / convertTransform vectors (round 4), on committed clouds
        make_score_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mt":          # cases added after the first fixtures were frozen
        make_mt_case("omp_direct7_mt", pair=15, n_az=128, n_beams=32, mode=DIRECT7, variant=0)
        make_mt_case("pca_direct1_mt", pair=17, n_az=128, n_beams=32, mode=DIRECT1, variant=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pcakd":
        make_case("pca_kdtree_r1", pair=21, n_az=128, n_beams=32, resolution=1.0, mode=KDTREE, variant=1, n_src_sweep=150, n_src_align=300)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "kdtree":      # only the case added after the first fixtures were frozen
        make_case("omp_kdtree_r1", pair=13, n_az=128, n_beams=32, resolution=1.0, mode=KDTREE, variant=0)
        sys.exit(0)
    se3_vectors()
    make_case("omp_direct7_r1", pair=3, n_az=128, n_beams=32, resolution=1.0, mode=DIRECT7, variant=0)
    make_case("omp_direct1_r1", pair=5, n_az=128, n_beams=32, resolution=1.0, mode=DIRECT1, variant=0)
    make_case("omp_direct26_r2", pair=7, n_az=128, n_beams=32, resolution=2.0, mode=DIRECT26, variant=0, n_src_sweep=120, n_src_align=250)
    make_case("pca_direct7_r1", pair=9, n_az=128, n_beams=32, resolution=1.0, mode=DIRECT7, variant=1)
    make_case("pca_direct1_r05", pair=11, n_az=256, n_beams=32, resolution=0.5, mode=DIRECT1, variant=1)
    make_case("omp_kdtree_r1", pair=13, n_az=128, n_beams=32, resolution=1.0, mode=KDTREE, variant=0)
    make_case("pca_kdtree_r1", pair=21, n_az=128, n_beams=32, resolution=1.0, mode=KDTREE, variant=1, n_src_sweep=150, n_src_align=300)
    make_score_cases()
