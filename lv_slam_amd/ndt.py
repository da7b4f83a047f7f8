"""Host-side mirror of the reference registration interface over the C-ABI (include/mi355_ndt.h).

`NormalDistributionsTransform` keeps the method names, argument meaning and error behaviour of
pclomp::/pclpca::NormalDistributionsTransform (include/ndt_omp/ndt_omp.h:69-277,
include/ndt_pca/ndt_pca.h) and of the pcl::Registration base it derives from, so a user of
lv_slam's `reg.setInputTarget(..); reg.setInputSource(..); reg.align(out, guess)` sequence
(src/lidar_odometry/scan_matching_odom_nodelet.cpp:109-119,197,220-226) finds the same surface.
All computation happens in libmi355ndt.so (hand-written HIP, gfx950).  There is no CPU fallback:
loading fails loudly if the library is missing, construction fails if no GPU is usable.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355NDT_LIB: another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("MI355NDT_LIB") or os.path.join(_HERE, "libmi355ndt.so")

# pclomp::NeighborSearchMethod (include/ndt_omp/ndt_omp.h:51-56)
KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3
VARIANT_OMP, VARIANT_PCA = 0, 1

OK = 0
ERR = {-1: "bad handle", -2: "bad argument", -3: "HIP error", -4: "target grid unusable", -5: "no HIP device",
       -6: "unsupported configuration", -7: "target/source not set"}


class NDTError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {ERR.get(code, code)}" + (f" ({detail})" if detail else ""))


class Params(C.Structure):
    _fields_ = [("resolution", C.c_float), ("step_size", C.c_double), ("outlier_ratio", C.c_double),
                ("trans_epsilon", C.c_double), ("max_iterations", C.c_int), ("neighbor_mode", C.c_int),
                ("variant", C.c_int), ("min_points_per_voxel", C.c_int), ("min_covar_eigvalue_mult", C.c_double)]


class Result(C.Structure):
    _fields_ = [("final_colmajor", C.c_float * 16), ("trans_probability", C.c_double), ("score", C.c_double),
                ("iterations", C.c_int), ("converged", C.c_int), ("sweeps", C.c_int), ("status", C.c_int),
                ("hits_last", C.c_longlong)]


class Voxel(C.Structure):
    _fields_ = [("idx", C.c_int32), ("n", C.c_int32), ("mean", C.c_double * 3), ("icov", C.c_float * 9),
                ("weight", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("sweep_launches", C.c_longlong), ("sweep_alg_bytes", C.c_double),
                ("sweep_hits", C.c_longlong), ("sweep_points", C.c_longlong), ("build_ms", C.c_double),
                ("build_launches", C.c_longlong), ("build_alg_bytes", C.c_double), ("update_ms", C.c_double),
                ("update_launches", C.c_longlong), ("async_fallbacks", C.c_longlong), ("stream_launches", C.c_longlong),
                ("stream_carried", C.c_longlong), ("stream_redone", C.c_longlong), ("cloud_uploads", C.c_longlong), ("cloud_upload_bytes", C.c_longlong),
                ("cloud_transfers", C.c_longlong), ("cloud_promotions", C.c_longlong),
                ("stream_reserved_slots", C.c_longlong), ("stream_launch_slots", C.c_longlong)]


class SeqParams(C.Structure):
    _fields_ = [("keyframe_delta_trans", C.c_double), ("keyframe_delta_angle", C.c_double), ("keyframe_delta_time", C.c_double)]


class SeqFrame(C.Structure):
    _fields_ = [("odom_colmajor", C.c_double * 16), ("tf_s2k_colmajor", C.c_float * 16), ("trans_probability", C.c_double),
                ("dx", C.c_double), ("da", C.c_double), ("dt", C.c_double), ("key_id", C.c_int), ("new_keyframe", C.c_int),
                ("iterations", C.c_int), ("converged", C.c_int), ("aligns", C.c_int), ("pad", C.c_int)]


class SeqStats(C.Structure):
    _fields_ = [("upload_ms", C.c_double), ("build_ms", C.c_double), ("track_ms", C.c_double), ("aligns", C.c_longlong),
                ("update_launches", C.c_longlong)]


# every symbol include/mi355_ndt.h declares
SYMBOLS = [
    "mi355ndt_version", "mi355ndt_device_count", "mi355ndt_host_numa_node", "mi355ndt_default_params", "mi355ndt_create", "mi355ndt_destroy",
    "mi355ndt_set_params", "mi355ndt_get_params", "mi355ndt_set_stream", "mi355ndt_last_error",
    "mi355ndt_set_target", "mi355ndt_set_source", "mi355ndt_promote_source_to_target", "mi355ndt_align", "mi355ndt_get_aligned", "mi355ndt_get_incremental",
    "mi355ndt_get_fitness_score", "mi355ndt_fitness_score_T", "mi355ndt_prefilter", "mi355ndt_use_prefiltered", "mi355ndt_derivatives", "mi355ndt_compute_hessian", "mi355ndt_derivatives_T", "mi355ndt_get_grid", "mi355ndt_get_voxels",
    "mi355ndt_batch_reserve", "mi355ndt_batch_set_target", "mi355ndt_batch_set_source", "mi355ndt_batch_set_clouds", "mi355ndt_batch_bind_device",
    "mi355ndt_batch_build_targets", "mi355ndt_batch_align", "mi355ndt_batch_size", "mi355ndt_batch_pose_records",
    "mi355ndt_profile_enable", "mi355ndt_profile_reset", "mi355ndt_profile_get", "mi355ndt_synchronize",
    "mi355ndt_set_latency_mode", "mi355ndt_sequence_run",
    "mi355ndt_calculate_score", "mi355ndt_convert_transform", "mi355ndt_set_option", "mi355ndt_get_option",
    "mi355ndt_stream_begin", "mi355ndt_stream_submit", "mi355ndt_stream_submit_host", "mi355ndt_stream_collect", "mi355ndt_stream_end", "mi355ndt_stream_pose_records", "mi355ndt_pack_pose_records",
]
OPT_ASYNC_ALIGN = 2            # mi355ndt_option: 1 (default) = one persistent launch per batch align, 0 = lockstep (update, sweep) rounds; same bits
OPT_DEBUG_ASYNC_ABORT = 3      # mi355ndt_option (test hook): the wave that claims this position of ring 0 gives up -> the batch is re-run in rounds
OPT_DEBUG_ASYNC_RINGS = 6      # mi355ndt_option (test hook): bit x clear -> ring x of a one-launch align has no workgroups of its own
OPT_STREAM_RESERVE = 5         # mi355ndt_option: workgroup slots the stream's launches leave free for the next batch's build (0 = off)
OPT_STREAM_THRESHOLD = 4       # mi355ndt_option: pairs a stream launch hands over to the next one (-1 = auto, 0 = none)
WARN_TOLERANCE_ARITH = 1       # mi355ndt_result.status under OPT_ARITH = 1: fewer than 4,096 hits at the final pose, or stopped at the iteration cap
OPT_ARITH = 7                  # mi355ndt_option: 0 = the reference recipe's arithmetic, one rounding per operation (default), 1 = tolerance arithmetic (held to 1e-4 m / 1e-5 rad, not to bits)
OPT_F32_SUM_ORDER = 1          # mi355ndt_option: 0 = (t0 + t1) + t2 (canonical), 1 = (t0 + t2) + t1 (Eigen 3.3 SSE predux pairing)

_LIB = None


def load_library(path: str = LIB_PATH):
    """dlopen libmi355ndt.so; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(path)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    L.mi355ndt_version.restype = C.c_char_p
    L.mi355ndt_last_error.restype = C.c_char_p
    L.mi355ndt_last_error.argtypes = [vp]
    L.mi355ndt_default_params.argtypes = [C.POINTER(Params)]
    L.mi355ndt_host_numa_node.argtypes = [i]
    L.mi355ndt_create.argtypes = [C.POINTER(Params), i, C.POINTER(vp)]
    L.mi355ndt_destroy.argtypes = [vp]
    L.mi355ndt_set_params.argtypes = [vp, C.POINTER(Params)]
    L.mi355ndt_get_params.argtypes = [vp, C.POINTER(Params)]
    L.mi355ndt_set_stream.argtypes = [vp, vp]
    L.mi355ndt_set_target.argtypes = [vp, vp, sz, sz]
    L.mi355ndt_set_source.argtypes = [vp, vp, sz, sz]
    L.mi355ndt_promote_source_to_target.argtypes = [vp]
    L.mi355ndt_align.argtypes = [vp, vp, C.POINTER(Result)]
    L.mi355ndt_get_aligned.argtypes = [vp, vp, sz]
    L.mi355ndt_get_incremental.argtypes = [vp, i, vp, vp]
    L.mi355ndt_get_fitness_score.argtypes = [vp, C.c_double, vp, vp]
    L.mi355ndt_fitness_score_T.argtypes = [vp, vp, C.c_double, vp, vp]
    L.mi355ndt_prefilter.argtypes = [vp, vp, sz, sz, i, C.c_double, C.c_double, C.c_float, vp, sz, sz, C.POINTER(sz)]
    L.mi355ndt_use_prefiltered.argtypes = [vp, i]
    L.mi355ndt_derivatives.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mi355ndt_derivatives_T.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.mi355ndt_compute_hessian.argtypes = [vp, vp, vp]
    L.mi355ndt_get_grid.argtypes = [vp, i, vp, vp, vp, vp]
    L.mi355ndt_get_voxels.argtypes = [vp, i, vp, sz]
    L.mi355ndt_batch_reserve.argtypes = [vp, i, sz, sz]
    L.mi355ndt_batch_set_target.argtypes = [vp, i, vp, sz, sz]
    L.mi355ndt_batch_set_source.argtypes = [vp, i, vp, sz, sz]
    L.mi355ndt_batch_set_clouds.argtypes = [vp, i, i, vp, vp, vp, vp, sz, i]
    L.mi355ndt_batch_bind_device.argtypes = [vp, i, vp, vp, sz, vp, vp, sz]
    L.mi355ndt_batch_build_targets.argtypes = [vp]
    L.mi355ndt_batch_align.argtypes = [vp, vp, vp]
    L.mi355ndt_batch_size.argtypes = [vp]
    L.mi355ndt_batch_pose_records.argtypes = [vp, i, i, vp, sz]
    L.mi355ndt_profile_enable.argtypes = [vp, i]
    L.mi355ndt_profile_reset.argtypes = [vp]
    L.mi355ndt_profile_get.argtypes = [vp, C.POINTER(Profile)]
    L.mi355ndt_synchronize.argtypes = [vp]
    L.mi355ndt_set_latency_mode.argtypes = [vp, i]
    L.mi355ndt_sequence_run.argtypes = [vp, i, vp, vp, sz, vp, C.POINTER(SeqParams), vp, vp, C.POINTER(SeqStats)]
    L.mi355ndt_calculate_score.argtypes = [vp, vp, sz, sz, C.POINTER(C.c_double)]
    L.mi355ndt_convert_transform.argtypes = [vp, vp]
    L.mi355ndt_set_option.argtypes = [vp, i, i]
    L.mi355ndt_get_option.argtypes = [vp, i, C.POINTER(i)]
    L.mi355ndt_stream_begin.argtypes = [vp, i, i, sz, sz]
    L.mi355ndt_stream_submit.argtypes = [vp, i, vp, vp, sz, vp, vp, sz, vp, C.POINTER(C.c_longlong)]
    L.mi355ndt_stream_submit_host.argtypes = [vp, i, vp, vp, vp, vp, sz, vp, i, C.POINTER(C.c_longlong)]
    L.mi355ndt_stream_collect.argtypes = [vp, C.c_longlong, vp]
    L.mi355ndt_stream_end.argtypes = [vp]
    L.mi355ndt_stream_pose_records.argtypes = [vp, vp, sz, i, i]
    L.mi355ndt_pack_pose_records.argtypes = [vp, i, i, i, vp, sz]
    _LIB = L
    return L


def default_params(**kw) -> Params:
    p = Params()
    load_library().mi355ndt_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _as_points(cloud) -> np.ndarray:
    """Accept [N,3]/[N,4+] float arrays (PCL-like records: x,y,z first); returns C-contiguous f32."""
    a = np.asarray(cloud)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be [N,>=3] (x,y,z first)")
    return np.ascontiguousarray(a, dtype=np.float32)


def _colmajor(M) -> np.ndarray:
    M = np.asarray(M, dtype=np.float32)
    if M.shape != (4, 4):
        raise ValueError("transform must be 4x4")
    return np.ascontiguousarray(M.T).ravel()      # column-major flattening (Eigen::Matrix4f layout)


def _result_dict(r: Result) -> dict:
    return dict(final=np.array(r.final_colmajor, np.float32).reshape(4, 4).T.copy(),
                trans_probability=r.trans_probability, score=r.score, iterations=r.iterations,
                converged=bool(r.converged), sweeps=r.sweeps, status=r.status, hits_last=r.hits_last)


class Engine:
    """Thin RAII wrapper of one mi355ndt_handle (one GPU, one stream)."""

    def __init__(self, params: Params | None = None, device: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.mi355ndt_create(C.byref(params) if params is not None else None, device, C.byref(self.h))
        if rc != OK:
            self.h = None
            raise NDTError(rc, "mi355ndt_create")
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.mi355ndt_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, where):
        if rc != OK:
            raise NDTError(rc, where, (self.lib.mi355ndt_last_error(self.h) or b"").decode())

    # -- parameters
    def get_params(self) -> Params:
        p = Params()
        self._chk(self.lib.mi355ndt_get_params(self.h, C.byref(p)), "get_params")
        return p

    def set_params(self, p: Params):
        self._chk(self.lib.mi355ndt_set_params(self.h, C.byref(p)), "set_params")

    def set_stream(self, stream_ptr: int | None):
        self._chk(self.lib.mi355ndt_set_stream(self.h, C.c_void_p(stream_ptr or 0)), "set_stream")

    # -- single registration
    def set_target(self, cloud):
        a = _as_points(cloud)
        self._chk(self.lib.mi355ndt_set_target(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0]), "set_target")

    def promote_source_to_target(self):
        """the cloud last set as source becomes the target, device to device (the nodelet's keyframe switch, scan_matching_odom_nodelet.cpp:240-243)"""
        self._chk(self.lib.mi355ndt_promote_source_to_target(self.h), "promote_source_to_target")

    def set_source(self, cloud):
        a = _as_points(cloud)
        self._n_src = a.shape[0]
        self._chk(self.lib.mi355ndt_set_source(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0]), "set_source")

    def align(self, guess) -> dict:
        g = _colmajor(guess)
        r = Result()
        self._chk(self.lib.mi355ndt_align(self.h, g.ctypes.data_as(C.c_void_p), C.byref(r)), "align")
        return _result_dict(r)

    def get_aligned(self) -> np.ndarray:
        out = np.zeros((self._n_src, 3), np.float32)
        self._chk(self.lib.mi355ndt_get_aligned(self.h, out.ctypes.data_as(C.c_void_p), 12), "get_aligned")
        return out

    def get_incremental(self, pair: int = 0):
        """(transformation_, previous_transformation_) of the last align: f32 exp(delta_p) of the last two Newton steps."""
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        self._chk(self.lib.mi355ndt_get_incremental(self.h, pair, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "get_incremental")
        return a.reshape(4, 4).T.copy(), b.reshape(4, 4).T.copy()

    def fitness_score(self, max_range: float = float("inf"), T=None):
        """getFitnessScore(max_range); T = explicit 4x4 transform (default: final pose of the last align)."""
        s, n = C.c_double(), C.c_longlong()
        mr = 1.7976931348623157e308 if max_range == float("inf") else float(max_range)
        if T is None:
            self._chk(self.lib.mi355ndt_get_fitness_score(self.h, mr, C.byref(s), C.byref(n)), "get_fitness_score")
        else:
            t = _colmajor(T)
            self._chk(self.lib.mi355ndt_fitness_score_T(self.h, t.ctypes.data_as(C.c_void_p), mr, C.byref(s), C.byref(n)), "fitness_score_T")
        return s.value, n.value

    def calculate_score(self, cloud) -> float:
        """calculateScore(cloud) (ndt_omp.h:232): negative log-likelihood of an ALREADY TRANSFORMED cloud against the target grid."""
        a = _as_points(cloud)
        s = C.c_double()
        self._chk(self.lib.mi355ndt_calculate_score(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0] if a.shape[0] else 12, C.byref(s)), "calculate_score")
        return s.value

    def set_option(self, option: int, value: int):
        self._chk(self.lib.mi355ndt_set_option(self.h, option, value), "set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int()
        self._chk(self.lib.mi355ndt_get_option(self.h, option, C.byref(v)), "get_option")
        return v.value

    def prefilter(self, cloud, distance_near=0.5, distance_far=100.0, downsample_resolution=0.1, use_distance_filter=True,
                  fetch=True):
        """PrefilteringNodelet distance_filter + VoxelGrid downsample (defaults of launch/dlo_kitti.launch:30-36).
        Returns the filtered [M,3] cloud (or just M when fetch=False; the result stays on the GPU for use_prefiltered)."""
        a = _as_points(cloud)
        n_out = C.c_size_t()
        out = np.zeros((a.shape[0], 3), np.float32) if fetch else None
        self._chk(self.lib.mi355ndt_prefilter(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0] if a.shape[0] else 12,
                                              int(use_distance_filter), float(distance_near), float(distance_far),
                                              float(downsample_resolution), out.ctypes.data_as(C.c_void_p) if fetch else None,
                                              a.shape[0], 12, C.byref(n_out)), "prefilter")
        self._pf_count = n_out.value
        return out[: n_out.value].copy() if fetch else n_out.value

    def use_prefiltered(self, as_target: bool):
        """setInputTarget / setInputSource with the last prefilter result, device to device."""
        self._chk(self.lib.mi355ndt_use_prefiltered(self.h, 2 if as_target else 1), "use_prefiltered")
        if not as_target:
            self._n_src = self._pf_count

    # -- parity hooks
    def derivatives(self, p):
        p = np.ascontiguousarray(p, np.float64)
        s, hits = C.c_double(), C.c_longlong()
        g, H = np.zeros(6), np.zeros(36)
        self._chk(self.lib.mi355ndt_derivatives(self.h, p.ctypes.data_as(C.c_void_p), C.byref(s), g.ctypes.data_as(C.c_void_p),
                                                H.ctypes.data_as(C.c_void_p), C.byref(hits)), "derivatives")
        return s.value, g, H.reshape(6, 6), hits.value

    def compute_hessian(self, p):
        """computeHessian (ndt_omp_impl2.hpp:622-679) at tangent p -> H[6,6] f64."""
        p = np.ascontiguousarray(p, np.float64)
        H = np.zeros(36)
        self._chk(self.lib.mi355ndt_compute_hessian(self.h, p.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)), "compute_hessian")
        return H.reshape(6, 6)

    def derivatives_T(self, T, Rj):
        t = _colmajor(T)
        r = np.ascontiguousarray(np.asarray(Rj, np.float32)).ravel()
        s, hits = C.c_double(), C.c_longlong()
        g, H = np.zeros(6), np.zeros(36)
        self._chk(self.lib.mi355ndt_derivatives_T(self.h, t.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.byref(s),
                                                  g.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p), C.byref(hits)), "derivatives_T")
        return s.value, g, H.reshape(6, 6), hits.value

    def get_grid(self, pair: int = 0):
        mn, mx, dv = (np.zeros(3, np.int32) for _ in range(3))
        nv = C.c_int()
        self._chk(self.lib.mi355ndt_get_grid(self.h, pair, mn.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p),
                                             dv.ctypes.data_as(C.c_void_p), C.byref(nv)), "get_grid")
        return mn, mx, dv, nv.value

    def get_voxels(self, pair: int = 0) -> np.ndarray:
        nv = self.get_grid(pair)[3]
        arr = (Voxel * max(nv, 1))()
        self._chk(self.lib.mi355ndt_get_voxels(self.h, pair, C.cast(arr, C.c_void_p), nv), "get_voxels")
        dt = np.dtype([("idx", "<i4"), ("n", "<i4"), ("mean", "<f8", 3), ("icov", "<f4", 9), ("weight", "<i4")], align=True)
        assert dt.itemsize == C.sizeof(Voxel)
        return np.frombuffer(bytes(arr), dtype=dt)[:nv].copy()

    # -- batch
    def batch_reserve(self, n_pairs, max_target_pts, max_source_pts):
        self._chk(self.lib.mi355ndt_batch_reserve(self.h, n_pairs, max_target_pts, max_source_pts), "batch_reserve")

    def batch_set_target(self, pair, cloud):
        a = _as_points(cloud)
        self._chk(self.lib.mi355ndt_batch_set_target(self.h, pair, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0]), "batch_set_target")

    def batch_set_source(self, pair, cloud):
        a = _as_points(cloud)
        self._chk(self.lib.mi355ndt_batch_set_source(self.h, pair, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0]), "batch_set_source")

    def batch_set_target_raw(self, pair: int, ptr: int, n: int, stride: int):
        """batch_set_target on a raw host pointer (records `stride` bytes apart); thread-safe across different pairs."""
        self._chk(self.lib.mi355ndt_batch_set_target(self.h, pair, C.c_void_p(ptr), n, stride), "batch_set_target")

    def batch_set_source_raw(self, pair: int, ptr: int, n: int, stride: int):
        self._chk(self.lib.mi355ndt_batch_set_source(self.h, pair, C.c_void_p(ptr), n, stride), "batch_set_source")

    def batch_set_clouds_raw(self, first_pair: int, tgt_ptrs, tgt_counts, src_ptrs, src_counts, stride: int, threads: int = 8):
        """mi355ndt_batch_set_clouds: arrays of host pointers / point counts (numpy uint64), one per pair; either side may be None."""
        n = len(tgt_ptrs if tgt_ptrs is not None else src_ptrs)
        arr = lambda a: None if a is None else np.ascontiguousarray(a, np.uint64)
        tp, tc, sp, sc = arr(tgt_ptrs), arr(tgt_counts), arr(src_ptrs), arr(src_counts)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.mi355ndt_batch_set_clouds(self.h, first_pair, n, ptr(tp), ptr(tc), ptr(sp), ptr(sc), stride, threads), "batch_set_clouds")

    def batch_bind_device(self, d_targets_ptr: int, target_counts, target_pitch: int, d_sources_ptr: int, source_counts, source_pitch: int):
        """Zero-copy device buffers laid out [pair][3][pitch] float32.  The caller keeps them alive."""
        tc = np.ascontiguousarray(target_counts, np.int32)
        sc = np.ascontiguousarray(source_counts, np.int32)
        assert len(tc) == len(sc)
        self._chk(self.lib.mi355ndt_batch_bind_device(self.h, len(tc), C.c_void_p(d_targets_ptr), tc.ctypes.data_as(C.c_void_p), target_pitch,
                                                      C.c_void_p(d_sources_ptr), sc.ctypes.data_as(C.c_void_p), source_pitch), "batch_bind_device")

    def batch_build_targets(self):
        self._chk(self.lib.mi355ndt_batch_build_targets(self.h), "batch_build_targets")

    def batch_align(self, guesses) -> list[dict]:
        G = np.asarray(guesses, np.float32)
        n = self.lib.mi355ndt_batch_size(self.h)
        if G.shape == (4, 4):
            G = np.broadcast_to(G, (n, 4, 4))
        if G.shape != (n, 4, 4):
            raise ValueError(f"guesses must be [{n},4,4]")
        gc = np.ascontiguousarray(np.transpose(G, (0, 2, 1))).reshape(n, 16)
        res = (Result * n)()
        self._chk(self.lib.mi355ndt_batch_align(self.h, gc.ctypes.data_as(C.c_void_p), C.cast(res, C.c_void_p)), "batch_align")
        return [_result_dict(r) for r in res]

    def batch_align_raw(self, guesses_colmajor: np.ndarray, res_array):
        """Allocation-free variant for timing loops: guesses [n,16] f32 C-contiguous, res_array = (Result*n)()."""
        self._chk(self.lib.mi355ndt_batch_align(self.h, guesses_colmajor.ctypes.data_as(C.c_void_p), C.cast(res_array, C.c_void_p)), "batch_align")

    def batch_pose_records(self, id_base: int, id_stride: int, d_records_ptr: int, capacity: int):
        """96-byte pose records of the last batch_align, packed on the device into a caller-owned device buffer (dist.py layout)."""
        self._chk(self.lib.mi355ndt_batch_pose_records(self.h, id_base, id_stride, C.c_void_p(d_records_ptr), capacity), "batch_pose_records")

    def synchronize(self):
        self._chk(self.lib.mi355ndt_synchronize(self.h), "synchronize")

    # -- stream mode: batches overlap on the GPU (mi355ndt_stream_*)
    def stream_begin(self, n_contexts: int, max_pairs: int, max_target_pts: int, max_source_pts: int):
        self._chk(self.lib.mi355ndt_stream_begin(self.h, n_contexts, max_pairs, max_target_pts, max_source_pts), "stream_begin")

    def stream_submit(self, d_targets_ptr: int, target_counts, target_pitch: int, d_sources_ptr: int, source_counts, source_pitch: int,
                      guesses_colmajor: np.ndarray) -> int:
        """Enqueue one batch (device-resident SoA buffers as in batch_bind_device; they must stay unchanged until the batch is collected).
        Returns the batch id at once."""
        tc = np.ascontiguousarray(target_counts, np.int32)
        sc = np.ascontiguousarray(source_counts, np.int32)
        g = np.ascontiguousarray(guesses_colmajor, np.float32)
        assert len(tc) == len(sc) and g.size == 16 * len(tc)
        bid = C.c_longlong(-1)
        self._chk(self.lib.mi355ndt_stream_submit(self.h, len(tc), C.c_void_p(d_targets_ptr), tc.ctypes.data_as(C.c_void_p), target_pitch,
                                                  C.c_void_p(d_sources_ptr), sc.ctypes.data_as(C.c_void_p), source_pitch,
                                                  g.ctypes.data_as(C.c_void_p), C.byref(bid)), "stream_submit")
        return bid.value

    def stream_submit_host_raw(self, tgt_ptrs, tgt_counts, src_ptrs, src_counts, stride: int, guesses_colmajor: np.ndarray, threads: int = 8) -> int:
        """Enqueue one batch of HOST clouds (arrays of record pointers and point counts, as batch_set_clouds_raw takes them); returns the batch id
        as soon as the caller's memory is no longer needed."""
        tp = np.ascontiguousarray(tgt_ptrs, np.uint64); sp = np.ascontiguousarray(src_ptrs, np.uint64)
        tc = np.ascontiguousarray(tgt_counts, np.uint64); sc = np.ascontiguousarray(src_counts, np.uint64)
        g = np.ascontiguousarray(guesses_colmajor, np.float32)
        assert len(tp) == len(sp) == len(tc) == len(sc) and g.size == 16 * len(tp)
        bid = C.c_longlong(-1)
        self._chk(self.lib.mi355ndt_stream_submit_host(self.h, len(tp), tp.ctypes.data_as(C.c_void_p), tc.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p),
                                                       sc.ctypes.data_as(C.c_void_p), C.c_size_t(stride), g.ctypes.data_as(C.c_void_p), int(threads), C.byref(bid)), "stream_submit_host")
        return bid.value

    def stream_pose_records(self, d_records_ptr: int, capacity: int, id_base: int, id_stride: int):
        """The NEXT submitted batch's 96-byte pose records go into this device buffer, written by the device as the pairs finish."""
        self._chk(self.lib.mi355ndt_stream_pose_records(self.h, C.c_void_p(d_records_ptr), capacity, id_base, id_stride), "stream_pose_records")

    def stream_collect_raw(self, batch_id: int, res_array):
        """Block until every pair of the batch is finalised; res_array = (Result * n_pairs)()."""
        self._chk(self.lib.mi355ndt_stream_collect(self.h, batch_id, C.cast(res_array, C.c_void_p)), "stream_collect")

    def stream_collect(self, batch_id: int, n_pairs: int) -> list[dict]:
        res = (Result * n_pairs)()
        self.stream_collect_raw(batch_id, res)
        return [_result_dict(r) for r in res]

    def stream_end(self):
        self._chk(self.lib.mi355ndt_stream_end(self.h), "stream_end")

    # -- latency mode
    def set_latency_mode(self, on: bool = True):
        """Opt-in fine-grained sweep for small batches (a single registration above all); see mi355ndt_set_latency_mode."""
        self._chk(self.lib.mi355ndt_set_latency_mode(self.h, int(on)), "set_latency_mode")

    def sequence_run(self, frames, stamps, keyframe_delta_trans=5.0, keyframe_delta_angle=0.17, keyframe_delta_time=1.0):
        """mi355ndt_sequence_run: a run of frames tracked on the device the way ScanMatchingOdomNodelet::matching_s2k tracks them
        (scan_matching_odom_nodelet.cpp:192-261).  frames: list of [N,>=3] float arrays (x,y,z first); stamps: seconds.
        Returns (list of per-frame dicts, stats dict)."""
        clouds = [_as_points(f) for f in frames]
        n = len(clouds)
        strides = {c.strides[0] for c in clouds if c.shape[0]} or {12}       # (numpy gives empty arrays zero strides)
        if len(strides) != 1:
            raise ValueError("all frames must share one record stride")
        ptrs = np.array([c.ctypes.data for c in clouds], np.uint64)
        cnts = np.array([c.shape[0] for c in clouds], np.uint64)
        st = np.ascontiguousarray(stamps, np.float64)
        if len(st) != n:
            raise ValueError("one stamp per frame")
        sp = SeqParams(keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time)
        out, res, stats = (SeqFrame * n)(), (Result * n)(), SeqStats()
        self._chk(self.lib.mi355ndt_sequence_run(self.h, n, ptrs.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p), strides.pop(),
                                                 st.ctypes.data_as(C.c_void_p), C.byref(sp), C.cast(out, C.c_void_p), C.cast(res, C.c_void_p),
                                                 C.byref(stats)), "sequence_run")
        frames_out = [dict(odom=np.array(f.odom_colmajor, np.float64).reshape(4, 4).T.copy(), tf_s2k=np.array(f.tf_s2k_colmajor, np.float32).reshape(4, 4).T.copy(),
                           trans_probability=f.trans_probability, test=(f.dx, f.da, f.dt), key_id=f.key_id, new_keyframe=bool(f.new_keyframe),
                           iterations=f.iterations, converged=bool(f.converged), aligns=f.aligns, result=_result_dict(r)) for f, r in zip(out, res)]
        return frames_out, {k: getattr(stats, k) for k, _ in SeqStats._fields_}

    # -- profiling
    def profile_enable(self, on=True):
        self._chk(self.lib.mi355ndt_profile_enable(self.h, int(on)), "profile_enable")

    def profile_reset(self):
        self._chk(self.lib.mi355ndt_profile_reset(self.h), "profile_reset")

    def profile_get(self) -> dict:
        p = Profile()
        self._chk(self.lib.mi355ndt_profile_get(self.h, C.byref(p)), "profile_get")
        return {k: getattr(p, k) for k, _ in Profile._fields_}


class NormalDistributionsTransform:
    """Drop-in mirror of pclomp::NormalDistributionsTransform / pclpca::NormalDistributionsTransform.

    variant=VARIANT_OMP -> include/ndt_omp/ndt_omp.h ; variant=VARIANT_PCA -> include/ndt_pca/ndt_pca.h.
    Method names (including the reference's `setOulierRatio` spelling) follow ndt_omp.h:109-203 and
    pcl::Registration.  PCL-style error behaviour: no exceptions for a failed registration --
    `hasConverged()` reports it; exceptions are reserved for API misuse and HIP failures.
    """

    def __init__(self, variant: int = VARIANT_OMP, device: int = 0):
        self._prm = default_params(variant=variant)           # ctor defaults, ndt_omp_impl2.hpp:53-83
        self._eng = Engine(self._prm, device)
        self._final = np.eye(4, dtype=np.float32)
        self._converged = False
        self._nr_iterations = 0
        self._trans_probability = 0.0
        self._has_target = False
        self._has_source = False
        self._last = None

    def _push(self):
        self._eng.set_params(self._prm)

    # ---- setters / getters of ndt_omp.h
    def setNumThreads(self, n: int):            # ndt_omp.h:109 -- OpenMP thread count; meaningless on the GPU, accepted
        self._num_threads = int(n)

    def setResolution(self, resolution: float):  # ndt_omp.h:126-136: re-voxelises only if changed AND a source is set (`if (input_) init();`), as the engine does
        if np.float32(resolution) != np.float32(self._prm.resolution):
            self._prm.resolution = float(resolution)
            self._push()

    def getResolution(self) -> float:
        return float(self._prm.resolution)

    def setStepSize(self, step_size: float):
        self._prm.step_size = float(step_size)
        self._push()

    def getStepSize(self) -> float:
        return self._prm.step_size

    def setOulierRatio(self, outlier_ratio: float):   # [sic] ndt_omp.h:181
        self._prm.outlier_ratio = float(outlier_ratio)
        self._push()

    def getOulierRatio(self) -> float:
        return self._prm.outlier_ratio

    def setNeighborhoodSearchMethod(self, method: int):
        self._prm.neighbor_mode = int(method)
        self._push()

    def setTransformationEpsilon(self, eps: float):    # pcl::Registration
        self._prm.trans_epsilon = float(eps)
        self._push()

    def setMaximumIterations(self, n: int):            # pcl::Registration
        self._prm.max_iterations = int(n)
        self._push()

    def setLatencyMode(self, on: bool = True):         # not in the reference: mi355ndt_set_latency_mode (opt-in fine-grained sweep)
        self._eng.set_latency_mode(on)

    def getTransformationProbability(self) -> float:
        return self._trans_probability

    def getTargetCells(self) -> np.ndarray:
        """pclpca getTargetCells() (ndt_pca.h:129-133): the searchable leaves of the target grid, in std::map order."""
        return self._eng.get_voxels(0)

    def getFinalNumIteration(self) -> int:
        return self._nr_iterations

    # ---- pcl::Registration surface
    def setInputTarget(self, cloud):             # ndt_omp.h:116-121 -> init()
        self._eng.set_target(cloud)
        self._has_target = True

    def setInputSource(self, cloud):
        self._eng.set_source(cloud)
        self._has_source = True

    def align(self, guess=None) -> np.ndarray:
        """align(output, guess): returns the output cloud (source moved by the final pose, [N,3] f32)."""
        if not (self._has_target and self._has_source):
            # pcl::Registration::initCompute() fails -> PCL prints an error and returns with converged_ unchanged
            self._converged = False
            return np.zeros((0, 3), np.float32)
        G = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, np.float32)
        r = self._eng.align(G)
        if r["status"] == WARN_TOLERANCE_ARITH:      # the tolerance arithmetic on a registration it is not meant for: once more in the default arithmetic
            self._eng.set_option(OPT_ARITH, 0)
            try:
                r = self._eng.align(G)
            finally:
                self._eng.set_option(OPT_ARITH, 1)
        self._last = r
        self._final = r["final"]
        self._converged = r["converged"]
        self._nr_iterations = r["iterations"]
        self._trans_probability = r["trans_probability"]
        return self._eng.get_aligned()

    def getFinalTransformation(self) -> np.ndarray:
        return self._final.copy()

    def calculateScore(self, cloud) -> float:
        """ndt_omp.h:232 / ndt_pca.h:244: negative log-likelihood of an already transformed cloud (lower is better)."""
        if not self._has_target:
            raise NDTError(-7, "calculateScore")
        return self._eng.calculate_score(cloud)

    @staticmethod
    def convertTransform(x) -> np.ndarray:
        """static convertTransform (ndt_omp.h:209-228): [x, y, z, roll, pitch, yaw] -> 4x4 f32 (Translation * Rx * Ry * Rz)."""
        v = np.ascontiguousarray(x, np.float64)
        if v.shape != (6,):
            raise ValueError("x must have six entries: x, y, z, roll, pitch, yaw")
        out = np.zeros(16, np.float32)
        rc = load_library().mi355ndt_convert_transform(v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise NDTError(rc, "convert_transform")
        return out.reshape(4, 4).T.copy()

    def setArithmetic(self, mode: int):                # not in the reference: mi355ndt_set_option(MI355NDT_OPT_ARITH)
        self._eng.set_option(OPT_ARITH, int(mode))

    def setF32SumOrder(self, order: int):              # not in the reference: mi355ndt_set_option(MI355NDT_OPT_F32_SUM_ORDER)
        self._eng.set_option(OPT_F32_SUM_ORDER, int(order))

    def getLastIncrementalTransformation(self) -> np.ndarray:
        """pcl::Registration::getLastIncrementalTransformation(): transformation_ = f32 exp(delta_p) of the last step (impl2:163)."""
        return self._eng.get_incremental(0)[0] if self._last is not None else np.eye(4, dtype=np.float32)

    def hasConverged(self) -> bool:
        return bool(self._converged)

    def getFitnessScore(self, max_range: float = float("inf")) -> float:
        """pcl::Registration::getFitnessScore (loop_detector.hpp:256): mean squared NN distance within max_range."""
        if not (self._has_target and self._has_source):
            return 1.7976931348623157e308
        return self._eng.fitness_score(max_range)[0]

    @property
    def engine(self) -> Engine:
        return self._eng
