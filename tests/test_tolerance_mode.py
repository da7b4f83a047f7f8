"""GPU (-m gpu): the tolerance arithmetic (MI355NDT_OPT_ARITH = 1; ndt_sweep.hpp eval_hit_fast, ndt_build.hpp k_leafsum_tree) held to what it promises --
north_star's SE(3) tolerance against the oracle (trans < 1e-4 m, rot < 1e-5 rad), the oracle's iteration counts -- and to what it keeps: the same leaves
for the same pose, determinism, a pair's bits independent of the batch it is in and of the path (rounds / one launch / stream) it takes.

The exact arithmetic stays the default and is what every other parity test runs; nothing here relaxes those.
(updateDerivatives: include/ndt_omp/ndt_omp_impl2.hpp:566-619; pca weighting ndt_pca_impl2.hpp:294-296; leaf sums voxel_grid_covariance_omp_impl.hpp:226-262.)"""
import numpy as np
import pytest

from conftest import se3_err
from lv_slam_amd import ndt, synth
from oracle import oracle_py as O
from test_gpu_configs import resident_batch

pytestmark = pytest.mark.gpu

CONFIGS = {
    "config3": (1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)),
    "nodelet": (1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=1)),
    "config5_d7": (2048, dict(resolution=0.5, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=1)),
    "config5_d1": (2048, dict(resolution=0.5, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=1)),
    "omp_d1": (1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=0)),
}


def engine(kw, arith, opts=None):
    e = ndt.Engine(ndt.default_params(**kw))
    e.set_option(ndt.OPT_ARITH, arith)
    for k, v in (opts or {}).items():
        e.set_option(k, v)
    assert e.get_option(ndt.OPT_ARITH) == arith
    return e


def words(res):
    return [(r["final"].tobytes(), r["score"], r["iterations"], r["converged"], r["sweeps"], r["hits_last"], r["trans_probability"]) for r in res]


@pytest.mark.parametrize("name,pair_ids,oracle_pairs", [("config3", list(range(24)), None), ("nodelet", list(range(24)), None),
                                                        ("config5_d7", list(range(128)), list(range(0, 128, 11))), ("config5_d1", list(range(128)), list(range(0, 128, 11))),
                                                        ("omp_d1", [3, 9, 200], None)])
def test_tolerance_mode_vs_oracle(name, pair_ids, oracle_pairs):
    """BASELINE configs 3 and 5 (and the nodelet's registration) under MI355NDT_OPT_ARITH = 1, through the engine's default path for that batch (24 x 65,536 points and
    128 x 131,072 points take the one-launch align): every checked pair has the oracle's iteration count, converged flag and hit count, and its pose inside the tolerance."""
    naz, kw = CONFIGS[name]
    T, S, host, n = resident_batch(pair_ids, naz)
    B = len(pair_ids)
    G = synth.default_guess()
    eng = engine(kw, 1)
    eng.batch_bind_device(T.data_ptr(), [n] * B, n, S.data_ptr(), [n] * B, n)
    eng.batch_build_targets()
    res = eng.batch_align(G)
    op = O.default_params(**kw)
    for k in (range(B) if oracle_pairs is None else oracle_pairs):
        tgt, src, _ = host[k]
        grid = O.Grid(tgt, op)
        # the grid: same leaves, same counts; means and inverse covariances from tree sums instead of ordered ones (1e-16 on the sums)
        lv = grid.leaves()
        sel = lv[(lv["n"] >= op.min_points_per_voxel) | (lv["n"] == -1)]
        v = eng.get_voxels(k)
        assert np.array_equal(v["idx"], sel["idx"]) and np.array_equal(v["n"], sel["n"])
        live = sel["n"] >= op.min_points_per_voxel
        assert np.allclose(v["mean"], sel["mean"], rtol=1e-13, atol=1e-13)
        scale = np.abs(sel["icov"][live]).max(axis=1, keepdims=True)
        assert np.all(np.abs(v["icov"][live] - sel["icov"][live].astype(np.float32)) <= 2e-6 * scale)
        if kw["variant"] == 1:
            assert np.array_equal(v["weight"][live], sel["weight"][live])
        ro = O.align(grid, src, G)
        r = res[k]
        assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"] and r["sweeps"] == ro["sweeps"], (k, r, ro)
        assert r["hits_last"] == ro["hits_last"]          # same leaves met at the final pose
        dt, dr = se3_err(ro["final"], r["final"])
        assert dt < 1e-4 and dr < 1e-5, (k, dt, dr)
        assert abs(r["score"] - ro["score"]) <= 1e-5 * max(1.0, abs(ro["score"]))
    res2 = eng.batch_align(G)                             # deterministic run to run
    assert words(res) == words(res2)
    eng.close()


@pytest.mark.parametrize("name", ["config3", "nodelet", "config5_d7"])
def test_one_sweep_in_tolerance_arithmetic_vs_oracle(name):
    """One computeDerivatives sweep (score, g, H) at a fixed tangent: 1e-5 of the largest entry (the exact arithmetic's bar is 1e-11), the oracle's hit count exactly."""
    naz, kw = CONFIGS[name]
    tgt, src, _ = synth.make_pair(2, naz)
    tgt, src = tgt.numpy(), src.numpy()
    p = np.array([0.9, 0.02, -0.01, 0.003, -0.002, 0.01])
    eng = engine(kw, 1)
    eng.set_target(tgt); eng.set_source(src)
    s, g, H, hits = eng.derivatives(p)
    op = O.default_params(**kw)
    so, go, Ho, ho = O.derivatives_at(O.Grid(tgt, op), src, p)
    assert hits == ho
    assert abs(s - so) <= 1e-5 * abs(so)
    assert np.max(np.abs(np.asarray(g) - go)) <= 1e-5 * np.max(np.abs(go))
    assert np.max(np.abs(np.asarray(H).reshape(6, 6) - np.asarray(Ho).reshape(6, 6))) <= 1e-5 * np.max(np.abs(Ho))
    eng.close()


@pytest.mark.parametrize("name", ["config3", "nodelet"])
def test_a_pairs_bits_do_not_depend_on_batch_or_path(name):
    """In the tolerance arithmetic too: the round-based align, the one-launch align (forced), a batch and a single registration give every pair the same result words."""
    naz, kw = CONFIGS[name]
    ids = [0, 4, 9, 13, 21]
    T, S, host, n = resident_batch(ids, naz)
    B = len(ids)
    G = synth.default_guess()
    out = {}
    for label, a_opt in (("rounds", 0), ("one_launch", 2)):
        eng = engine(kw, 1, {ndt.OPT_ASYNC_ALIGN: a_opt})
        eng.batch_bind_device(T.data_ptr(), [n] * B, n, S.data_ptr(), [n] * B, n)
        eng.batch_build_targets()
        eng.profile_enable(True); eng.profile_reset()
        out[label] = eng.batch_align(G)
        pr = eng.profile_get()
        assert (pr["update_launches"] == 0) == (label == "one_launch"), pr
        eng.close()
    assert words(out["rounds"]) == words(out["one_launch"])
    eng = engine(kw, 1)
    for k in (1, 3):
        eng.set_target(host[k][0]); eng.set_source(host[k][1])
        r = eng.align(G)
        assert words([r]) == words([out["rounds"][k]])
    eng.close()


def test_stream_in_tolerance_arithmetic_equals_the_synchronous_align():
    naz, kw = CONFIGS["config3"]
    nb, nbat = 20, 3
    T, S, host, n = resident_batch(list(range(nb * nbat)), naz)
    G = synth.default_guess()
    guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (nb, 16)), dtype=np.float32)
    fsz = 4 * 3 * n
    eng = engine(kw, 1, {ndt.OPT_ASYNC_ALIGN: 2})
    ref = []
    for b in range(nbat):
        eng.batch_bind_device(T.data_ptr() + b * nb * fsz, [n] * nb, n, S.data_ptr() + b * nb * fsz, [n] * nb, n)
        eng.batch_build_targets()
        r = (ndt.Result * nb)()
        eng.batch_align_raw(guesses, r)
        ref.append(bytes(r))
    eng.stream_begin(3, nb, n, n)
    # (what a stream computes with is fixed at stream_begin: the arithmetic cannot change under it)
    with pytest.raises(ndt.NDTError):
        eng.set_option(ndt.OPT_ARITH, 0)
    with pytest.raises(ndt.NDTError):
        eng.set_option(ndt.OPT_F32_SUM_ORDER, 1)
    ids = [eng.stream_submit(T.data_ptr() + (i % nbat) * nb * fsz, [n] * nb, n, S.data_ptr() + (i % nbat) * nb * fsz, [n] * nb, n, guesses) for i in range(2)]
    for i in range(2, 7):
        r = (ndt.Result * nb)()
        eng.stream_collect_raw(ids[i - 2], r)
        assert bytes(r) == ref[(i - 2) % nbat], i
        ids.append(eng.stream_submit(T.data_ptr() + (i % nbat) * nb * fsz, [n] * nb, n, S.data_ptr() + (i % nbat) * nb * fsz, [n] * nb, n, guesses))
    for i in (5, 6):
        r = (ndt.Result * nb)()
        eng.stream_collect_raw(ids[i], r)
        assert bytes(r) == ref[i % nbat], i
    eng.stream_end()
    eng.close()


@pytest.mark.parametrize("kw_extra", [dict(neighbor_mode=ndt.DIRECT26), dict(neighbor_mode=ndt.KDTREE), dict(neighbor_mode=ndt.DIRECT7, step_size=0.004)])
def test_configurations_the_tolerance_arithmetic_does_not_serve_keep_the_exact_kernels(kw_extra):
    """DIRECT26 / KDTREE and the live More-Thuente case (step_size <= eps / 2, ndt_omp_impl2.hpp:888) ignore the option: same result words as with it off."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=30, variant=0)
    kw.update(kw_extra)
    tgt, src, _ = synth.make_pair(1, 256, n_beams=32)
    tgt, src = tgt.numpy(), src.numpy()
    G = synth.default_guess()
    r = []
    for arith in (0, 1):
        eng = engine(kw, arith)
        eng.set_target(tgt); eng.set_source(src)
        r.append(eng.align(G))
        eng.close()
    assert words([r[0]]) == words([r[1]])


def test_switching_the_arithmetic_rebuilds_what_it_needs():
    """Grids built under one arithmetic lack the other's records: the option may be flipped at any time between aligns and the next align answers as a fresh engine would."""
    naz, kw = CONFIGS["config3"]
    tgt, src, _ = synth.make_pair(5, naz)
    tgt, src = tgt.numpy(), src.numpy()
    G = synth.default_guess()
    fresh = {}
    for arith in (0, 1):
        eng = engine(kw, arith)
        eng.set_target(tgt); eng.set_source(src)
        fresh[arith] = words([eng.align(G)])
        eng.close()
    assert fresh[0] != fresh[1]                       # (the two arithmetics do differ in the last bits of this pose)
    eng = engine(kw, 0)
    eng.set_target(tgt); eng.set_source(src)
    assert words([eng.align(G)]) == fresh[0]
    eng.set_option(ndt.OPT_ARITH, 1)
    assert words([eng.align(G)]) == fresh[1]
    eng.set_option(ndt.OPT_ARITH, 0)
    assert words([eng.align(G)]) == fresh[0]
    eng.close()


def test_latency_mode_in_tolerance_arithmetic_vs_oracle():
    """The fine-grained single-registration sweep (mi355ndt_set_latency_mode) has tolerance-arithmetic instantiations too."""
    naz, kw = CONFIGS["nodelet"]
    tgt, src, _ = synth.make_pair(7, naz)
    tgt, src = tgt.numpy(), src.numpy()
    G = synth.default_guess()
    eng = engine(kw, 1)
    eng.set_latency_mode(True)
    eng.set_target(tgt); eng.set_source(src)
    r = eng.align(G)
    ro = O.align(O.Grid(tgt, O.default_params(**kw)), src, G)
    assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"]
    dt, dr = se3_err(ro["final"], r["final"])
    assert dt < 1e-4 and dr < 1e-5, (dt, dr)
    eng.close()


def test_a_registration_the_mode_is_not_meant_for_is_flagged_and_the_mirror_reruns_it():
    """Fewer than MI355NDT_TOLERANCE_MIN_HITS hits at the final pose (a source of a few hundred points: the Hessian of a handful of leaves) or a stop at the
    iteration cap: the result carries MI355NDT_WARN_TOLERANCE_ARITH in its status -- under the tolerance arithmetic only, for the pair alone, by every path --
    and the reference-shaped mirror (like the pcl adaptor) answers with the default arithmetic's result instead."""
    naz, kw = CONFIGS["nodelet"]
    tgt, src, _ = synth.make_pair(4, naz)
    tgt, src = tgt.numpy(), src.numpy()
    few = np.ascontiguousarray(src[::211])            # ~310 points
    G = synth.default_guess()
    res = {}
    for arith in (0, 1):
        eng = engine(kw, arith)
        eng.set_target(tgt)
        eng.set_source(few)
        res[arith] = eng.align(G)
        eng.set_source(src)
        assert eng.align(G)["status"] == 0            # a full scan: no caveat in either arithmetic
        # the same two pairs as a batch: the flag is the pair's, not the batch's
        eng.batch_reserve(2, len(tgt), len(src))
        for b, s in enumerate((few, src)):
            eng.batch_set_target(b, tgt); eng.batch_set_source(b, s)
        eng.batch_build_targets()
        rb = eng.batch_align(G)
        assert [r["status"] for r in rb] == [arith, 0] and words([rb[0]]) == words([res[arith]])
        eng.close()
    assert res[0]["status"] == 0 and res[1]["status"] == ndt.WARN_TOLERANCE_ARITH and res[1]["hits_last"] < 4096
    # a run that stops at the iteration cap
    eng = engine(dict(kw, max_iterations=1), 1)
    eng.set_target(tgt); eng.set_source(src)
    r = eng.align(G)
    assert r["iterations"] == 3 and r["status"] == ndt.WARN_TOLERANCE_ARITH
    eng.close()
    # the mirror: setArithmetic(1), few-hit source -> the default arithmetic's words
    reg = ndt.NormalDistributionsTransform(variant=ndt.VARIANT_PCA)
    reg.setResolution(1.0); reg.setNeighborhoodSearchMethod(ndt.DIRECT1); reg.setTransformationEpsilon(0.01); reg.setMaximumIterations(64)
    reg.setArithmetic(1)
    reg.setInputTarget(tgt); reg.setInputSource(few)
    reg.align(G)
    assert np.array_equal(reg.getFinalTransformation(), res[0]["final"]) and reg.getFinalNumIteration() == res[0]["iterations"]
    assert reg.engine.get_option(ndt.OPT_ARITH) == 1
    reg.setInputSource(src)
    reg.align(G)                                       # ... and a full scan stays in the tolerance arithmetic
    eng = engine(kw, 1)
    eng.set_target(tgt); eng.set_source(src)
    assert np.array_equal(reg.getFinalTransformation(), eng.align(G)["final"])
    eng.close(); reg.engine.close()
