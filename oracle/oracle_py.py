"""ctypes binding of oracle/libndt_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from lv_slam_amd/.  See oracle/ndt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3
VARIANT_OMP, VARIANT_PCA = 0, 1


class Params(C.Structure):
    _fields_ = [("resolution", C.c_float), ("step_size", C.c_double), ("outlier_ratio", C.c_double),
                ("trans_epsilon", C.c_double), ("max_iterations", C.c_int), ("neighbor_mode", C.c_int),
                ("variant", C.c_int), ("min_points_per_voxel", C.c_int), ("min_covar_eigvalue_mult", C.c_double)]


class Leaf(C.Structure):
    _fields_ = [("idx", C.c_int32), ("n", C.c_int32), ("mean", C.c_double * 3), ("cov", C.c_double * 9),
                ("icov", C.c_double * 9), ("evals", C.c_double * 3), ("evecs", C.c_double * 9),
                ("label", C.c_int32), ("weight", C.c_int32), ("dim2d", C.c_double), ("centroid", C.c_float * 3),
                ("n_pushed", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("final_colmajor", C.c_float * 16), ("trans_probability", C.c_double), ("score", C.c_double),
                ("iterations", C.c_int), ("converged", C.c_int), ("hits_last", C.c_long), ("sweeps", C.c_int), ("mt_loops", C.c_int),
                ("inc_colmajor", C.c_float * 16), ("prev_inc_colmajor", C.c_float * 16)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libndt_oracle.so")
    src = os.path.join(_HERE, "ndt_oracle.c")
    inc = os.path.join(_HERE, "ndt_oracle_refshape.inc")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(inc)):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.ora_grid_build.restype = C.c_void_p
        L.ora_grid_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Params)]
        L.ora_grid_free.argtypes = [C.c_void_p]
        L.ora_grid_num_leaves.restype = C.c_size_t
        L.ora_grid_num_leaves.argtypes = [C.c_void_p]
        L.ora_grid_num_valid.restype = C.c_size_t
        L.ora_grid_num_valid.argtypes = [C.c_void_p]
        L.ora_grid_leaves.restype = C.POINTER(Leaf)
        L.ora_grid_leaves.argtypes = [C.c_void_p]
        L.ora_grid_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_derivatives.restype = C.c_long
        L.ora_derivatives.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_derivatives_at.restype = C.c_long
        L.ora_derivatives_at.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_align.restype = C.c_int
        L.ora_align.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                C.c_void_p, C.POINTER(Result)]
        L.ora_compute_hessian.restype = None
        L.ora_compute_hessian.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_calculate_score.restype = C.c_double
        L.ora_calculate_score.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ora_convert_transform.restype = None
        L.ora_convert_transform.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_se3_log.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_se3_compose_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_svd_solve6.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_eigen_sym3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_gauss_constants.argtypes = [C.c_double, C.c_float, C.c_void_p]
        L.ora_default_params.argtypes = [C.POINTER(Params)]
        L.ora_set_threads.argtypes = [C.c_int]
        L.ora_refgrid_build.restype = C.c_void_p
        L.ora_refgrid_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Params)]
        L.ora_refgrid_free.argtypes = [C.c_void_p]
        L.ora_refgrid_num_leaves.restype = C.c_size_t
        L.ora_refgrid_num_leaves.argtypes = [C.c_void_p]
        L.ora_refgrid_leaves.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_ref_align.restype = C.c_int
        L.ora_ref_align.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Result)]
        L.ora_exp_f32arg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ora_prefilter.restype = C.c_size_t
        L.ora_prefilter.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_float] + [C.c_void_p] * 3
        L.ora_policy_step.restype = C.c_int
        L.ora_policy_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_fitness_score.restype = C.c_double
        L.ora_fitness_score.argtypes = [C.c_void_p] * 3 + [C.c_size_t] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def default_params(**kw) -> Params:
    p = Params()
    lib().ora_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _soa(pts: np.ndarray):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    return (np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1]), np.ascontiguousarray(pts[:, 2]))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Grid:
    """VoxelGridCovariance restatement (voxel_grid_covariance_omp_impl.hpp:48-370)."""

    def __init__(self, pts: np.ndarray, prm: Params):
        self.prm = prm
        self._xyz = _soa(pts)
        self.h = lib().ora_grid_build(_p(self._xyz[0]), _p(self._xyz[1]), _p(self._xyz[2]), len(pts), C.byref(prm))

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.ora_grid_free(self.h)
            self.h = None

    @property
    def ok(self):
        return bool(self.h)

    def num_leaves(self):
        return lib().ora_grid_num_leaves(self.h)

    def num_valid(self):
        return lib().ora_grid_num_valid(self.h)

    def bounds(self):
        a, b, c = (np.zeros(3, np.int32) for _ in range(3))
        lib().ora_grid_bounds(self.h, _p(a), _p(b), _p(c))
        return a, b, c

    def leaves(self):
        """dict of numpy arrays over ALL leaves in ascending idx order."""
        n = self.num_leaves()
        arr = lib().ora_grid_leaves(self.h)
        buf = np.ctypeslib.as_array(C.cast(arr, C.POINTER(C.c_uint8)), shape=(n * C.sizeof(Leaf),))
        dt = np.dtype([("idx", "<i4"), ("n", "<i4"), ("mean", "<f8", 3), ("cov", "<f8", 9), ("icov", "<f8", 9),
                       ("evals", "<f8", 3), ("evecs", "<f8", 9), ("label", "<i4"), ("weight", "<i4"), ("dim2d", "<f8"),
                       ("centroid", "<f4", 3), ("n_pushed", "<i4")])
        assert dt.itemsize == C.sizeof(Leaf)
        return np.frombuffer(buf.tobytes(), dtype=dt)

    def valid_leaves(self):
        lv = self.leaves()
        return lv[lv["n"] >= self.prm.min_points_per_voxel]


def derivatives(grid: Grid, src: np.ndarray, T: np.ndarray, Rj: np.ndarray):
    """One sweep; T = 4x4 f32 (row-indexed numpy), Rj = 3x3 f32.  Returns (score, g[6], H[6,6], hits)."""
    x, y, z = _soa(src)
    Tc = np.asfortranarray(np.asarray(T, np.float32)).ravel(order="F").copy()
    Rr = np.ascontiguousarray(np.asarray(Rj, np.float32)).ravel().copy()
    s = C.c_double()
    g = np.zeros(6)
    H = np.zeros(36)
    hits = lib().ora_derivatives(grid.h, C.byref(grid.prm), _p(x), _p(y), _p(z), len(src), _p(Tc), _p(Rr),
                                 C.byref(s), _p(g), _p(H))
    return s.value, g, H.reshape(6, 6), hits


def derivatives_at(grid: Grid, src: np.ndarray, p: np.ndarray):
    x, y, z = _soa(src)
    p = np.ascontiguousarray(p, np.float64)
    s = C.c_double()
    g = np.zeros(6)
    H = np.zeros(36)
    hits = lib().ora_derivatives_at(grid.h, C.byref(grid.prm), _p(x), _p(y), _p(z), len(src), _p(p),
                                    C.byref(s), _p(g), _p(H))
    return s.value, g, H.reshape(6, 6), hits


def compute_hessian(grid: Grid, src: np.ndarray, p: np.ndarray):
    """ora_compute_hessian at tangent p (cloud moved by f32(exp(p))).  Returns H[6,6] f64."""
    x, y, z = _soa(src)
    p = np.ascontiguousarray(p, np.float64)
    M = se3_exp(p)
    Tc = np.asarray(M, np.float32).ravel(order="F").copy()
    H = np.zeros(36)
    lib().ora_compute_hessian(grid.h, C.byref(grid.prm), _p(x), _p(y), _p(z), len(src), _p(Tc), _p(p), _p(H))
    return H.reshape(6, 6)


def calculate_score(grid: Grid, cloud: np.ndarray, gauss=None):
    """calculateScore(cloud) (impl2:1006-1040): `cloud` is the ALREADY TRANSFORMED cloud; gauss = (d1, d2, d3) as the members hold them
    (default: those of the grid's parameters, i.e. what an align() with them leaves behind)."""
    x, y, z = _soa(cloud)
    g = np.ascontiguousarray(gauss_constants(grid.prm.outlier_ratio, grid.prm.resolution) if gauss is None else gauss, np.float64)
    return float(lib().ora_calculate_score(grid.h, _p(g), grid.prm.resolution, _p(x), _p(y), _p(z), len(cloud)))


def convert_transform(x) -> np.ndarray:
    """static convertTransform (ndt_omp.h:209-228): [x, y, z, roll, pitch, yaw] -> 4x4 f32."""
    v = np.ascontiguousarray(x, np.float64)
    out = np.zeros(16, np.float32)
    lib().ora_convert_transform(_p(v), _p(out))
    return out.reshape(4, 4, order="F").copy()


def align(grid: Grid, src: np.ndarray, guess: np.ndarray):
    """ora_align; guess = 4x4 f32.  Returns dict(final[4,4] f32, trans_probability, score, iterations, converged, ...)."""
    x, y, z = _soa(src)
    Gc = np.asarray(guess, np.float32).ravel(order="F").copy()
    r = Result()
    rc = lib().ora_align(grid.h, C.byref(grid.prm), _p(x), _p(y), _p(z), len(src), _p(Gc), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"ora_align rc={rc}")
    return dict(final=np.array(r.final_colmajor, np.float32).reshape(4, 4, order="F"),
                trans_probability=r.trans_probability, score=r.score, iterations=r.iterations,
                converged=bool(r.converged), hits_last=r.hits_last, sweeps=r.sweeps, mt_loops=r.mt_loops,
                transformation=np.array(r.inc_colmajor, np.float32).reshape(4, 4, order="F"),
                previous_transformation=np.array(r.prev_inc_colmajor, np.float32).reshape(4, 4, order="F"))


def se3_exp(p):
    p = np.ascontiguousarray(p, np.float64)
    M = np.zeros(16)
    lib().ora_se3_exp(_p(p), _p(M))
    return M.reshape(4, 4)


def se3_log(M):
    M = np.ascontiguousarray(M, np.float64).ravel().copy()
    p = np.zeros(6)
    lib().ora_se3_log(_p(M), _p(p))
    return p


def se3_compose_log(dp, p):
    dp = np.ascontiguousarray(dp, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    o = np.zeros(6)
    lib().ora_se3_compose_log(_p(dp), _p(p), _p(o))
    return o


def svd_solve6(H, b):
    H = np.ascontiguousarray(H, np.float64).ravel().copy()
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(6)
    lib().ora_svd_solve6(_p(H), _p(b), _p(x))
    return x


def eigen_sym3(A):
    A = np.ascontiguousarray(A, np.float64).ravel().copy()
    ev = np.zeros(3)
    V = np.zeros(9)
    lib().ora_eigen_sym3(_p(A), _p(ev), _p(V))
    return ev, V.reshape(3, 3)


def gauss_constants(outlier_ratio: float, resolution: float):
    o = np.zeros(3)
    lib().ora_gauss_constants(outlier_ratio, resolution, _p(o))
    return o


def fitness_score(tgt: np.ndarray, src: np.ndarray, T: np.ndarray, max_range: float = float("inf")):
    """getFitnessScore restatement: (score, inliers).  T = 4x4 f32."""
    tx, ty, tz = _soa(tgt)
    sx, sy, sz = _soa(src)
    Tc = np.asarray(T, np.float32).ravel(order="F").copy()
    n = C.c_long()
    mr = 1.7976931348623157e308 if max_range == float("inf") else float(max_range)
    s = lib().ora_fitness_score(_p(tx), _p(ty), _p(tz), len(tgt), _p(sx), _p(sy), _p(sz), len(src), _p(Tc), mr, C.byref(n))
    return s, n.value


def prefilter(pts: np.ndarray, distance_near=0.5, distance_far=100.0, leaf=0.1, use_distance_filter=True) -> np.ndarray:
    """distance_filter + VoxelGrid downsample restatement; returns [M,3] f32."""
    x, y, z = _soa(pts)
    n = len(pts)
    ox, oy, oz = (np.zeros(max(n, 1), np.float32) for _ in range(3))
    m = lib().ora_prefilter(_p(x), _p(y), _p(z), n, int(use_distance_filter), float(distance_near), float(distance_far), float(leaf),
                            _p(ox), _p(oy), _p(oz))
    return np.stack([ox[:m], oy[:m], oz[:m]], axis=1)


def sequence(frames, stamps, prm: Params, keyframe_delta_trans=5.0, keyframe_delta_angle=0.17, keyframe_delta_time=1.0):
    """ScanMatchingOdomNodelet::matching_s2k over a run of frames (scan_matching_odom_nodelet.cpp:192-261) with the oracle as the
    registration: frame 0 becomes the keyframe, frame 1 is aligned twice (:223-227), ora_policy_step does :229-250.
    Returns a list of dicts per frame: odom[4,4] f64, tf_s2k[4,4] f32, key_id, new_keyframe, iterations, converged, test (dx, da, dt)."""
    pre, key_pose = np.eye(4).ravel().copy(), np.eye(4).ravel().copy()
    kstamp = C.c_double(float(stamps[0]))
    thr = np.array([keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time], np.float64)
    out = [dict(odom=np.eye(4), tf_s2k=np.eye(4, dtype=np.float32), key_id=0, new_keyframe=True, iterations=0, converged=True, test=(0.0, 0.0, 0.0), aligns=0)]
    key_id, grid = 0, Grid(np.asarray(frames[0], np.float32), prm)
    guess = np.eye(4, dtype=np.float32)
    guess[0, 3] = 1.5                                                  # :199-200
    for k in range(1, len(frames)):
        cloud = np.asarray(frames[k], np.float32)
        r = align(grid, cloud, guess)
        n_al = 1
        if k == 1:                                                      # :223-227
            r = align(grid, cloud, r["final"])
            n_al = 2
        fin = np.asarray(r["final"], np.float32).ravel(order="F").copy()
        odom, g64, test = np.zeros(16), np.zeros(16), np.zeros(3)
        matched = key_id
        newkey = lib().ora_policy_step(_p(pre), _p(key_pose), C.byref(kstamp), _p(fin), float(stamps[k]), _p(thr), _p(odom), _p(g64), _p(test))
        if newkey:
            key_id, grid = k, Grid(cloud, prm)                          # :243 setInputTarget(key)
        guess = g64.reshape(4, 4).astype(np.float32)                    # guess_trans.cast<float>() (:221)
        out.append(dict(odom=odom.reshape(4, 4).copy(), tf_s2k=np.asarray(r["final"], np.float32), key_id=matched, new_keyframe=bool(newkey),
                        iterations=r["iterations"], converged=r["converged"], test=tuple(test), aligns=n_al, trans_probability=r["trans_probability"]))
    return out


class RefGrid:
    """The reference-shaped voxel grid (std::map-like tree, serial build) -- CPU-baseline timing only."""

    def __init__(self, target: np.ndarray, prm: Params):
        self.prm = prm
        x, y, z = _soa(target)
        self.h = lib().ora_refgrid_build(_p(x), _p(y), _p(z), len(target), C.byref(prm))

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.ora_refgrid_free(self.h)
            self.h = None

    def leaves(self):
        n = lib().ora_refgrid_num_leaves(self.h)
        arr = (Leaf * max(n, 1))()
        lib().ora_refgrid_leaves(self.h, C.cast(arr, C.c_void_p))
        dt = np.dtype([("idx", "<i4"), ("n", "<i4"), ("mean", "<f8", 3), ("cov", "<f8", 9), ("icov", "<f8", 9),
                       ("evals", "<f8", 3), ("evecs", "<f8", 9), ("label", "<i4"), ("weight", "<i4"), ("dim2d", "<f8"),
                       ("centroid", "<f4", 3), ("n_pushed", "<i4")])
        return np.frombuffer(bytes(arr), dtype=dt)[:n].copy()


def ref_align(grid: RefGrid, src: np.ndarray, guess: np.ndarray):
    x, y, z = _soa(src)
    Gc = np.asarray(guess, np.float32).ravel(order="F").copy()
    r = Result()
    rc = lib().ora_ref_align(grid.h, C.byref(grid.prm), _p(x), _p(y), _p(z), len(src), _p(Gc), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"ora_ref_align rc={rc}")
    return dict(final=np.array(r.final_colmajor, np.float32).reshape(4, 4, order="F"), trans_probability=r.trans_probability,
                score=r.score, iterations=r.iterations, converged=bool(r.converged), hits_last=r.hits_last, sweeps=r.sweeps)
