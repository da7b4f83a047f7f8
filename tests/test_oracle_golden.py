"""CPU: pin the C oracle (oracle/ndt_oracle.c) against the committed fixtures produced by the
independent NumPy restatement tests/golden/make_golden.py (different SE3/eigen/SVD algorithms,
literal matrix products).  Tolerances: voxel statistics rtol 1e-9, sweep (score,g,H) rtol 1e-9
(eigh-vs-Jacobi icov differences), align pose within the north-star SE(3) tolerance."""
import os
import numpy as np
import pytest

from oracle import oracle_py as O
from conftest import se3_err

CASES = ["omp_direct7_r1", "omp_direct1_r1", "omp_direct26_r2", "pca_direct7_r1", "pca_direct1_r05", "omp_kdtree_r1", "pca_kdtree_r1"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    pv = z["params"]
    prm = O.default_params(resolution=float(pv[0]), step_size=float(pv[1]), outlier_ratio=float(pv[2]),
                           trans_epsilon=float(pv[3]), max_iterations=int(pv[4]), neighbor_mode=int(pv[5]),
                           variant=int(pv[6]), min_points_per_voxel=int(pv[7]), min_covar_eigvalue_mult=float(pv[8]))
    return z, prm


def test_gauss_constants_analytic():
    # SURVEY.md A9 anchors
    d = O.gauss_constants(0.55, 1.0)
    assert np.allclose(d, [-2.21723, 0.43312, 0.59784], atol=2e-5)
    d = O.gauss_constants(0.55, 0.5)
    assert np.allclose(d, [-0.70445, 0.75636, -1.48160], atol=2e-5)


def test_se3_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "se3_vectors.npz"))
    for p, M, dp, cl in zip(z["p"], z["M"], z["dp"], z["compose_log"]):
        Mo = O.se3_exp(p)
        # Sophus' closed form (1-cos t)/t^2 cancels catastrophically for t ~ 1e-6 (SMALL_EPS is 1e-10):
        # ~1e-16/t^2 relative error in V, i.e. <= ~1e-9 m here.  That is the reference's behaviour, kept.
        tol = 1e-12 if np.linalg.norm(p[3:]) > 1e-3 or np.linalg.norm(p[3:]) == 0 else 1e-9
        assert np.allclose(Mo, M, atol=tol, rtol=1e-12)
        if np.linalg.norm(p[3:]) < 3.0:        # log is principal-branch; fixtures stay below pi
            assert np.allclose(O.se3_log(M), p, atol=1e-9)
        assert np.allclose(O.se3_compose_log(dp, p), cl, atol=2e-9)


def test_svd_solve_semantics():
    rng = np.random.default_rng(0)
    H = rng.normal(size=(6, 6))
    b = rng.normal(size=6)
    assert np.allclose(O.svd_solve6(H, b), np.linalg.solve(H, b), rtol=1e-9, atol=1e-12)
    # rank-deficient: min-norm solution (P14)
    H2 = H.copy()
    H2[:, 5] = H2[:, 0] * 2 - H2[:, 3]
    assert np.allclose(O.svd_solve6(H2, b), np.linalg.pinv(H2, rcond=6 * np.finfo(float).eps) @ b, rtol=1e-6, atol=1e-9)
    # H = 0 -> exactly 0
    assert np.all(O.svd_solve6(np.zeros((6, 6)), b) == 0)


def test_svd_solve_is_scale_invariant():
    """Eigen 3.3's JacobiSVD works on matrix / max|coeff| (JacobiSVD.h, compute(): `scale`), so the solve neither overflows nor
    underflows: ndt_pca's weights compound multiplicatively over the DIRECT26 neighbours (ndt_pca_impl2.hpp:294-296) and put
    Hessian entries at 1e87 and beyond (found by tools/fuzz_parity.py: seed 6, case 109)."""
    rng = np.random.default_rng(5)
    for _ in range(20):
        H = rng.normal(size=(6, 6))
        H = H + H.T                                             # symmetric, indefinite: what the Newton step sees
        b = rng.normal(size=6)
        x1 = np.linalg.solve(H, b)
        for k in (1e-250, 1e-150, 1e-90, 1.0, 1e90, 1e150, 1e250):
            xs = O.svd_solve6(H * k, b * k)
            assert np.allclose(xs, x1, rtol=1e-8, atol=1e-11), (k, xs, x1)
        # rank-deficient at a huge magnitude: still the minimum-norm solution
        H2 = H.copy()
        H2[:, 5] = H2[:, 0] * 2 - H2[:, 3]
        x2 = np.linalg.pinv(H2, rcond=6 * np.finfo(float).eps) @ b
        assert np.allclose(O.svd_solve6(H2 * 1e120, b * 1e120), x2, rtol=1e-6, atol=1e-9)


def test_eigen_sym3():
    rng = np.random.default_rng(1)
    for _ in range(50):
        A = rng.normal(size=(3, 3))
        A = A @ A.T * rng.uniform(1e-4, 10)
        ev, V = O.eigen_sym3(A)
        ev2 = np.linalg.eigvalsh(A)
        assert np.allclose(ev, ev2, rtol=1e-12, atol=1e-14)
        assert np.allclose(V @ np.diag(ev) @ V.T, A, atol=1e-12)
    # reads the lower triangle only
    A = np.array([[2.0, 99, 99], [0.5, 3.0, 99], [0.1, 0.2, 1.0]])
    L = np.tril(A) + np.tril(A, -1).T
    assert np.allclose(O.eigen_sym3(A)[0], np.linalg.eigvalsh(L))


@pytest.mark.parametrize("name", CASES)
def test_voxel_grid(golden_dir, name):
    z, prm = load(golden_dir, name)
    g = O.Grid(z["target"], prm)
    assert g.ok
    mn, mx, dv = g.bounds()
    assert np.array_equal(mn, z["min_b"]) and np.array_equal(mx, z["max_b"]) and np.array_equal(dv, z["div_b"])
    lv = g.leaves()
    assert np.array_equal(lv["idx"], z["leaf_idx"])
    assert np.array_equal(lv["n"], z["leaf_n"])
    assert np.allclose(lv["mean"], z["leaf_mean"], rtol=1e-14, atol=0)
    v = lv[z["v_sel"]]
    assert np.allclose(v["cov"].reshape(-1, 3, 3), z["v_cov"], rtol=1e-9, atol=1e-12)
    assert np.allclose(v["icov"].reshape(-1, 3, 3), z["v_icov"], rtol=1e-8, atol=1e-9)
    assert np.allclose(v["evals"], z["v_evals"], rtol=1e-9, atol=1e-13)
    assert np.array_equal(v["label"], z["v_label"])
    assert np.array_equal(v["weight"], z["v_weight"])
    assert np.allclose(O.gauss_constants(prm.outlier_ratio, prm.resolution), z["gauss"], rtol=1e-14)


@pytest.mark.parametrize("name", CASES)
def test_sweep(golden_dir, name):
    z, prm = load(golden_dir, name)
    g = O.Grid(z["target"], prm)
    for p, T, sc, gr, H, hits in zip(z["sweep_p"], z["sweep_T"], z["sweep_score"], z["sweep_g"], z["sweep_H"], z["sweep_hits"]):
        s, g6, H66, h = O.derivatives(g, z["src_sweep"], T, T[:3, :3])
        assert h == hits
        scale = max(1.0, np.abs(H).max())
        assert abs(s - sc) <= 1e-9 * max(1.0, abs(sc))
        assert np.allclose(g6, gr, rtol=1e-8, atol=1e-9 * scale)
        assert np.allclose(H66, H, rtol=1e-8, atol=1e-9 * scale)
        # the restated H is NOT symmetric -- make sure we kept that
        if hits > 0:
            assert np.abs(H66 - H66.T).max() > 0


@pytest.mark.parametrize("name", CASES)
def test_align(golden_dir, name):
    z, prm = load(golden_dir, name)
    g = O.Grid(z["target"], prm)
    r = O.align(g, z["src_align"], z["guess"])
    assert r["iterations"] == int(z["align_iterations"])
    assert r["converged"] == bool(z["align_converged"])
    dt, dr = se3_err(z["align_final"], r["final"])
    assert dt < 1e-4 and dr < 1e-5, (dt, dr)      # north-star SE(3) tolerance
    assert abs(r["score"] - float(z["align_score"])) <= 1e-6 * max(1.0, abs(float(z["align_score"])))
    assert r["sweeps"] == len(z["align_trace_score"])


MT_CASES = ["omp_direct7_mt", "pca_direct1_mt"]


@pytest.mark.parametrize("name", MT_CASES)
def test_compute_hessian(golden_dir, name):
    """computeHessian/updateHessian (impl2:622-714): f64 pass over kd-tree neighbourhoods, vs the NumPy restatement."""
    z, prm = load(golden_dir, name)
    grid = O.Grid(z["target"], prm)
    H = O.compute_hessian(grid, z["src_align"], z["hess_p"])
    assert np.abs(H - z["hess_H"]).max() <= 1e-11 * np.abs(z["hess_H"]).max()


@pytest.mark.parametrize("name", MT_CASES)
@pytest.mark.parametrize("tag", ["far", "near", "over"])
def test_align_live_more_thuente(golden_dir, name, tag):
    """step_size <= eps/2 makes computeStepLengthMT's loop and computeHessian live (impl2:888, 920-1000)."""
    z, prm = load(golden_dir, name)
    assert not (prm.step_size - prm.trans_epsilon / 2 > 0)
    grid = O.Grid(z["target"], prm)
    r = O.align(grid, z["src_align"], z["guess_" + tag])
    assert r["iterations"] == int(z[tag + "_iterations"]) and r["converged"] == bool(z[tag + "_converged"])
    assert r["mt_loops"] == int(z[tag + "_mt_its"].sum())
    dt, dr = se3_err(z[tag + "_final"], r["final"])
    assert dt < 1e-6 and dr < 1e-6, (dt, dr)         # f32 pose entries may differ by an ulp between expm and the closed form
    assert abs(r["score"] - float(z[tag + "_score"])) <= 1e-6 * abs(float(z[tag + "_score"]))


def test_more_thuente_scalar_pieces():
    """trialValueSelectionMT / updateIntervalMT / std::min,max NaN semantics, against the NumPy restatement."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    assert np.isnan(M.cmin(float("nan"), 1.0)) and M.cmin(1.0, float("nan")) == 1.0
    assert np.isnan(M.cmax(float("nan"), 1.0)) and M.cmax(1.0, float("nan")) == 1.0
    # interval update: U1, U2, U3, converged
    I = [0.0, 0.0, -1.0, 0.0, 0.0, -1.0]
    assert M.update_interval_mt(I, 0.5, 0.1, 0.2) is False and I[3:] == [0.5, 0.1, 0.2]
    I = [0.0, 0.0, -1.0, 0.0, 0.0, -1.0]
    assert M.update_interval_mt(I, 0.5, -0.1, -0.2) is False and I[:3] == [0.5, -0.1, -0.2]
    assert M.update_interval_mt(I, 0.5, -0.1, -0.2) is True                 # same point again: g_t * (a_l - a_t) == 0
    I = [0.0, 0.0, -1.0, 1.0, 0.3, 0.4]
    assert M.update_interval_mt(I, 0.5, -0.1, 0.2) is False and I == [0.5, -0.1, 0.2, 0.0, 0.0, -1.0]
    # cubic/quadratic trial of case 1 on phi(a) = (a-1)^2 - 1 sampled at 0 and 3: minimiser of both interpolants is a = 1
    a = M.trial_value_selection_mt(0.0, 0.0, -2.0, 0.0, 0.0, -2.0, 3.0, 3.0, 4.0)
    assert abs(a - 1.0) < 1e-12


def test_gradient_and_hessian_are_derivatives_of_the_score():
    """Analytic anchor: at p = 0 the reference's left-perturbation forms are plain derivatives, so the oracle's g must
    be d(score)/d(delta) and H(:, k) must be dg/d(delta_k) -- checked by central differences on a smooth case (one voxel, DIRECT1,
    source points that stay inside the cell, so no neighbour set changes under the perturbation)."""
    rng = np.random.default_rng(5)
    c = np.array([10.5, -7.5, 2.5])
    A = rng.normal(0, 1, (3, 3))
    tgt = (c + (rng.normal(0, 1, (400, 3)) * [0.12, 0.07, 0.03]) @ A.T * 0.3).astype(np.float32)
    tgt = tgt[(np.abs(tgt - c) < 0.49).all(1)]
    tgt = np.concatenate([tgt, np.array([[0.5, 0.5, 0.5], [20.5, -15.5, 5.5]], np.float32)])      # gives the grid some extent
    grid = O.Grid(tgt, O.default_params(neighbor_mode=O.DIRECT1))
    assert len(grid.valid_leaves()) == 1
    src = (c + rng.uniform(-0.25, 0.25, (300, 3))).astype(np.float32)
    s0, g0, H0, h0 = O.derivatives_at(grid, src, np.zeros(6))
    assert h0 == len(src)
    h = 1e-4
    gfd, Hfd = np.zeros(6), np.zeros((6, 6))
    for k in range(6):
        e = np.zeros(6)
        e[k] = h
        sp, gp, _, hp = O.derivatives_at(grid, src, e)
        sm, gm, _, hm = O.derivatives_at(grid, src, -e)
        assert hp == hm == h0
        gfd[k] = (sp - sm) / (2 * h)
        Hfd[:, k] = (gp - gm) / (2 * h)
    assert np.abs(gfd - g0).max() <= 1e-3 * np.abs(g0).max()
    assert np.abs(Hfd - H0).max() <= 1e-3 * np.abs(H0).max()
    # the Hessian is NOT symmetric (the point-Hessian term is asymmetric in the rotation block)
    assert np.abs(H0[3:, 3:] - H0[3:, 3:].T).max() > 1e-6 * np.abs(H0).max()


def test_align_edge_cases():
    prm = O.default_params(trans_epsilon=0.01, max_iterations=64)
    rng = np.random.default_rng(3)
    tgt = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
    g = O.Grid(tgt, prm)
    # source far away from every voxel: zero hits -> delta_p == 0 -> converged, 0 iterations, final == guess
    src = (rng.uniform(-1, 1, (100, 3)) + 500).astype(np.float32)
    G = np.eye(4, dtype=np.float32)
    r = O.align(g, src, G)
    assert r["converged"] and r["iterations"] == 0 and r["sweeps"] == 1 and np.array_equal(r["final"], G)
    # step_size <= eps/2 enables the More-Thuente loop; with zero hits it is never reached (delta_p == 0 returns first)
    prm2 = O.default_params(trans_epsilon=0.5, step_size=0.1)
    r2 = O.align(O.Grid(tgt, prm2), src, G)
    assert r2["converged"] and r2["iterations"] == 0 and r2["mt_loops"] == 0
    # non-finite points are skipped
    tgt2 = tgt.copy()
    tgt2[::7] = np.nan
    g2 = O.Grid(tgt2, prm)
    assert g2.ok and g2.leaves()["n"][g2.leaves()["n"] > 0].sum() == np.isfinite(tgt2).all(1).sum()


def test_fitness_score_vs_kdtree():
    """pins ora_fitness_score (brute force, f32 distances) against scipy's exact kd-tree in f64."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(9)
    tgt = rng.uniform(-20, 20, (5000, 3)).astype(np.float32)
    src = rng.uniform(-25, 25, (1500, 3)).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.3, -0.2, 0.1]
    c, s = np.float32(np.cos(0.05)), np.float32(np.sin(0.05))
    T[:2, :2] = [[c, -s], [s, c]]
    q = (src.astype(np.float64) @ T[:3, :3].astype(np.float64).T) + T[:3, 3].astype(np.float64)
    d, _ = cKDTree(tgt.astype(np.float64)).query(q)
    for mr in (0.5, 4.0, 25.0, float("inf")):
        sel = d * d <= mr
        exp = (d[sel] ** 2).mean() if sel.any() else 1.7976931348623157e308
        got, n = O.fitness_score(tgt, src, T, mr)
        assert abs(n - int(sel.sum())) <= 2            # f32 vs f64 at the threshold
        assert abs(got - exp) <= 2e-5 * max(1.0, exp)
    got, n = O.fitness_score(tgt, src + 1000.0, T, 1.0)
    assert n == 0 and got == 1.7976931348623157e308


def test_prefilter_vs_numpy():
    """pins ora_prefilter against a dictionary-based NumPy restatement of pcl::VoxelGrid + the distance filter."""
    rng = np.random.default_rng(21)
    pts = (rng.normal(0, 1, (20000, 3)) * np.array([30, 30, 2])).astype(np.float32)
    pts[::97] = np.nan
    pts[5] = [0.1, 0.1, 0.1]                 # inside the near threshold
    pts[6] = [150, 0, 0]                     # beyond the far threshold
    near, far, leaf = 0.5, 100.0, np.float32(0.1)
    d = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2]).astype(np.float64)
    keep = (d > near) & (d < far) & np.isfinite(pts).all(1)
    f = pts[keep]
    inv = np.float32(1.0) / leaf
    min_b = np.floor(f.min(0) * inv).astype(np.int64)
    max_b = np.floor(f.max(0) * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(f * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    cells = {}
    for k, p in zip(idx, f):
        s = cells.setdefault(int(k), [np.zeros(3, np.float32), 0])
        s[0] = (s[0] + p).astype(np.float32)
        s[1] += 1
    exp = np.array([cells[k][0] / np.float32(cells[k][1]) for k in sorted(cells)], np.float32)
    got = O.prefilter(pts, near, far, float(leaf))
    assert got.shape == exp.shape and np.array_equal(got, exp)
    # no down-sampling: the distance filter alone keeps the input order
    assert np.array_equal(O.prefilter(pts, near, far, 0.0), f)
    # leaf far too small for the extent: PCL warns and returns the (filtered) input
    assert np.array_equal(O.prefilter(pts, near, far, 1e-4), f)
    assert len(O.prefilter(np.zeros((0, 3), np.float32))) == 0


def test_svd_solve_non_finite_input_gives_nan():
    """Eigen 3.3 JacobiSVD keeps rank 6 on NaN singular values, solve() propagates NaN, and computeTransformation reports
    converged_ = false (ndt_omp_impl2.hpp:147-151); a thresholded pseudo-inverse must not answer 0 (= "converged") there."""
    H = np.eye(6)
    b = np.ones(6)
    for bad in (np.nan, np.inf, -np.inf):
        Hb = H.copy(); Hb[2, 3] = bad
        assert np.isnan(O.svd_solve6(Hb, b)).all()
        bb = b.copy(); bb[4] = bad
        assert np.isnan(O.svd_solve6(H, bb)).all()
    assert np.array_equal(O.svd_solve6(np.zeros((6, 6)), np.zeros(6)), np.zeros(6))      # H = 0, g = 0 stays the exact-zero case (P14)


def test_align_reports_incremental_transforms():
    """transformation_ = float(exp(delta_p)) of the last step, previous_transformation_ = the one before (impl2:134, 163)."""
    from lv_slam_amd import synth
    tgt, src, dT = synth.make_pair(2, 128, n_beams=32)
    tgt, src = tgt.numpy(), src.numpy()
    prm = O.default_params(trans_epsilon=0.01, max_iterations=64)
    r = O.align(O.Grid(tgt, prm), src, synth.default_guess())
    assert r["iterations"] >= 2
    for M in (r["transformation"], r["previous_transformation"]):
        assert np.allclose(M[3], [0, 0, 0, 1]) and np.allclose(M[:3, :3] @ M[:3, :3].T, np.eye(3), atol=1e-6)
    # the last step is the shorter one (the loop ends when |a| < eps)
    step = lambda M: np.linalg.norm(M[:3, 3])
    assert step(r["transformation"]) < 0.011 and step(r["transformation"]) <= step(r["previous_transformation"]) + 1e-9


@pytest.mark.parametrize("kw", [dict(), dict(variant=1, neighbor_mode=3), dict(variant=1, resolution=0.5), dict(neighbor_mode=1, resolution=2.0)])
def test_reference_shaped_mode_equals_the_port(kw):
    """oracle/ndt_oracle_refshape.inc (ordered-map grid, per-evaluation exp(p), dense 4x6 / 24x6 matrices, cloud rewrite per
    sweep, schedule(guided, 8)) is the arrangement the CPU baseline is timed on; it must compute what the port computes: the
    same leaves bit for bit, and the same alignment up to the f64 summation order of the threads."""
    from lv_slam_amd import synth
    tgt, src, dT = synth.make_pair(4, 128, n_beams=32)
    tgt, src = tgt.numpy(), src.numpy()
    prm = O.default_params(trans_epsilon=0.01, max_iterations=64, **kw)
    g, rg = O.Grid(tgt, prm), O.RefGrid(tgt, prm)
    a, b = g.leaves(), rg.leaves()
    for f in ("idx", "n", "mean", "cov", "icov", "evals", "weight", "centroid", "n_pushed"):
        assert np.array_equal(a[f], b[f], equal_nan=True), f
    G = synth.default_guess()
    for th in (1, 3):
        O.lib().ora_set_threads(th)
        r1, r2 = O.align(g, src, G), O.ref_align(rg, src, G)
        assert r1["iterations"] == r2["iterations"] and r1["converged"] == r2["converged"] and r1["hits_last"] == r2["hits_last"]
        dt, dr = se3_err(r1["final"], r2["final"])
        assert dt < 1e-6 and dr < 1e-7
        assert abs(r1["score"] - r2["score"]) <= 1e-9 * max(1.0, abs(r1["score"]))
    O.lib().ora_set_threads(0)


def test_leaf_too_small_guard_does_not_overflow():
    """voxel_grid_covariance_omp_impl.hpp:75-84 / pcl::VoxelGrid: dx*dy*dz > INT32_MAX -> no grid / cloud not down-sampled.  The
    reference's int64 product itself overflows beyond 2^63 cells (undefined behaviour); the restatement evaluates the guard without
    overflowing (found by tools/fuzz_parity.py: a 1e-4 m leaf over a 240 m cloud gave 1.5e19 cells, the wrapped product was negative
    and the points were binned with wrapped int32 indices)."""
    rng = np.random.default_rng(9)
    pts = rng.uniform(-120, 120, (4000, 3)).astype(np.float32)
    assert np.array_equal(O.prefilter(pts, 0.5, 1000.0, 1e-4, False), pts)            # 2.4e6 cells per axis: output = input, input order
    assert np.array_equal(O.prefilter(pts, 0.5, 100.0, 1e-4, True), O.prefilter(pts, 0.5, 100.0, 0.0, True))   # = the distance gate alone
    # the NDT target grid: a stray point at 1e30 (the f32 extent in cells is beyond 2^63), and a plain too-fine grid
    stray = pts.copy()
    stray[7] = [1e30, 0.0, 0.0]
    assert not O.Grid(stray, O.default_params()).ok
    assert not O.Grid(pts, O.default_params(resolution=1e-4)).ok
    assert O.Grid(pts, O.default_params(resolution=1.0)).ok


def test_calculate_score_vs_numpy_restatement(golden_dir):
    """calculateScore (ndt_omp_impl2.hpp:1006-1040 / ndt_pca_impl2.hpp:1013-1047): the oracle against the literal NumPy restatement, with
    the constructor's Gauss constants (what the members hold before any align, impl2:70-76) and with those of the fixture's parameters."""
    z = np.load(os.path.join(golden_dir, "calc_score.npz"))
    for tag in ("omp", "pca"):
        fx, prm = load(golden_dir, str(z[f"{tag}_fixture"]))
        grid = O.Grid(fx["target"], prm)
        assert np.allclose(O.gauss_constants(0.55, 1.0), z[f"{tag}_gauss_ctor"], rtol=1e-14)
        for kind, cloud in (("ctor", z[f"{tag}_cloud"]), ("align", z[f"{tag}_cloud"])):
            got = O.calculate_score(grid, cloud, z[f"{tag}_gauss_{kind}"])
            want = float(z[f"{tag}_score_{kind}"])
            assert abs(got - want) <= 1e-9 * abs(want), (tag, kind, got, want)       # eigh-vs-Jacobi icov differences, as for the sweeps
        got = O.calculate_score(grid, fx["src_align"])                                # default constants = the grid's parameters
        assert abs(got - float(z[f"{tag}_score_raw_source"])) <= 1e-9 * abs(float(z[f"{tag}_score_raw_source"]))
    # a cloud nowhere near the grid: every neighbourhood is empty, the score is exactly 0; an empty cloud: 0 / 0
    fx, prm = load(golden_dir, "omp_direct7_r1")
    grid = O.Grid(fx["target"], prm)
    assert O.calculate_score(grid, fx["src_align"] + np.float32(1e4)) == 0.0
    assert np.isnan(O.calculate_score(grid, np.zeros((0, 3), np.float32)))


def test_convert_transform_vs_scipy(golden_dir):
    """static convertTransform (ndt_omp.h:209-228): Translation * Rx(roll) * Ry(pitch) * Rz(yaw) in f32; fixture = scipy rotations in f64."""
    z = np.load(os.path.join(golden_dir, "calc_score.npz"))
    for x, M in zip(z["ct_x"], z["ct_M"]):
        got = O.convert_transform(x)
        assert got.dtype == np.float32 and np.abs(got.astype(np.float64) - M).max() < 1e-6
        assert np.array_equal(got[3], [0, 0, 0, 1]) and np.array_equal(got[:3, 3], x[:3].astype(np.float32))
    assert np.array_equal(O.convert_transform(np.zeros(6)), np.eye(4, dtype=np.float32))
