"""The update kernel's Newton solve (ndtm::lu_solve6 / ndtm::svd_solve6, lv_slam_amd/csrc/ndt_math.hpp) on the device against
numpy and against the oracle's ora_svd_solve6, system by system: `JacobiSVD(H).solve(-g)` of ndt_omp_impl2.hpp:138-140 --
well-conditioned, rank-deficient, zero, and at magnitudes from 1e-250 to 1e250 (Eigen 3.3's JacobiSVD scales by the largest
coefficient; ndt_pca's compounded weights reach 1e87 in practice, tools/fuzz_parity.py)."""
import os
import subprocess
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALES = (1e-250, 1e-150, 1e-90, 1e-20, 1.0, 1e20, 1e90, 1e150, 1e250)


def test_device_newton_solve_against_numpy_and_oracle(tmp_path):
    import __graft_entry__ as entry
    from oracle import oracle_py as O
    exe = entry.build_solve6_check()
    rng = np.random.default_rng(11)
    systems, kinds = [], []
    for _ in range(60):
        H = rng.normal(size=(6, 6)); H = H + H.T                 # symmetric indefinite, cond ~ 10..1000
        b = rng.normal(size=6)
        Hd = H.copy(); Hd[:, 5] = Hd[:, 0] * 2 - Hd[:, 3]; Hd[5, :] = Hd[0, :] * 2 - Hd[3, :]   # rank 5, still symmetric
        u = rng.normal(size=6)
        H1 = np.outer(u, u)                                      # rank 1: what a one-point source gives
        for k in SCALES:
            systems += [np.concatenate([(H * k).ravel(), b * k]), np.concatenate([(Hd * k).ravel(), b * k]), np.concatenate([(H1 * k).ravel(), b * k])]
            kinds += ["full", "rank5", "rank1"]
    # full rank but badly conditioned (cond 1e6 .. 1e12): the reference's route, not LU
    for c in (1e6, 1e8, 1e10, 1e12):
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        H = (Q * np.array([1.0, 0.5, 0.1, 0.05, 0.01, 1.0 / c])) @ Q.T
        systems.append(np.concatenate([H.ravel(), rng.normal(size=6)])); kinds.append("illcond")
    systems.append(np.zeros(42)); kinds.append("zero")
    A = np.ascontiguousarray(np.stack(systems))
    A.tofile(tmp_path / "in.f64")
    subprocess.check_call([exe, str(tmp_path / "in.f64"), str(tmp_path / "out.f64")], timeout=120)
    out = np.fromfile(tmp_path / "out.f64", np.float64).reshape(-1, 20)
    assert len(out) == len(A)
    n_lu = 0
    for row, o, kind in zip(A, out, kinds):
        H, b = row[:36].reshape(6, 6), row[36:]
        xs, xr, lu, xw = o[:6], o[6:12], bool(o[12]), o[14:20]
        n_lu += lu
        assert np.array_equal(xw, xr, equal_nan=True), (kind, xw, xr)      # seven lanes side by side == one lane: same bits, same route
        if kind == "zero":
            assert not lu and np.array_equal(xs, np.zeros(6)) and np.array_equal(xr, np.zeros(6))
            continue
        s = np.abs(H).max()
        want = np.linalg.pinv(H / s, rcond=6 * np.finfo(float).eps) @ (b / s)      # (numpy's own SVD is scale-safe only up to a point)
        loose = 1e-4 if kind == "illcond" else 1e-7                                  # (cond * eps)
        tol = dict(rtol=loose, atol=loose * 1e-2 * max(1.0, np.abs(want).max()))
        assert np.allclose(xs, want, **tol), (kind, s, xs, want)
        assert np.allclose(xr, want, **tol), (kind, s, lu, xr, want)
        xo = O.svd_solve6(H, b)                                                     # the oracle's restatement: same algorithm, same bits
        assert np.array_equal(xs, xo), (kind, s, xs, xo)
        if kind != "full":
            assert not lu, (kind, s)                                                # rank-deficient / ill-conditioned systems never take the LU route
            assert np.array_equal(xr, xo)
        else:
            # LU is accepted where (||H||_F ||H^-1||_F) < 1e5 and the squares it is computed from stay inside the f64 range
            condF = np.linalg.norm(H / s) * np.linalg.norm(np.linalg.inv(H / s))
            if 1e-140 < s < 1e140 and not (0.9e5 < condF < 1.1e5):
                assert lu == (condF < 1e5), (s, condF, lu)
            if not (1e-160 < s < 1e160):
                assert not lu, s
    assert n_lu > 60 * 5 * 0.9                                                      # (the random symmetric systems are almost all well conditioned)
