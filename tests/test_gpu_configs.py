"""GPU (-m gpu): the BASELINE.json configurations at their full sizes, through mi355ndt_batch_bind_device (device-resident
SoA buffers, the path bench.py times), every pair checked against the oracle: voxel grids bit-exact, same iteration count and
converged flag, SE(3) inside the north-star tolerance (trans < 1e-4 m, rot < 1e-5 rad).

  config 3 shape : a batch of 65,536-pt pairs, ndt_omp, 1 m, DIRECT7
  config 5       : 131,072-pt pairs, ndt_pca, 0.5 m, DIRECT7 and DIRECT1 (DIRECT1 = what the live nodelet sets,
                   scan_matching_odom_nodelet.cpp:109-119; weighting ndt_pca_impl2.hpp:294-296; gate ndt_omp_impl2.hpp:581-589)
"""
import numpy as np
import pytest

from conftest import se3_err
from lv_slam_amd import ndt, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def resident_batch(pair_ids, naz):
    import torch
    dev = torch.device("cuda:0")
    n = naz * 64
    T = torch.empty(len(pair_ids), 3, n, device=dev)
    S = torch.empty(len(pair_ids), 3, n, device=dev)
    host = []
    for k, pid in enumerate(pair_ids):
        t, s, dT = synth.make_pair(pid, naz, device=dev)
        T[k] = t.T
        S[k] = s.T
        host.append((t.cpu().numpy(), s.cpu().numpy(), dT))
    torch.cuda.synchronize()
    return T, S, host, n


def run_and_check(pair_ids, naz, kw, check_motion=True, oracle_pairs=None, one_launch=None):
    """oracle_pairs: batch slots held against the oracle (default: all); one_launch: assert that the align took (True) / did not take (False)
    the one-launch path of ndt_async.hpp -- the DEFAULT choice of the engine for that batch, nothing forced."""
    T, S, host, n = resident_batch(pair_ids, naz)
    B = len(pair_ids)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.batch_bind_device(T.data_ptr(), [n] * B, n, S.data_ptr(), [n] * B, n)
    eng.batch_build_targets()
    G = synth.default_guess()
    eng.profile_enable(True); eng.profile_reset()
    res = eng.batch_align(G)
    pr = eng.profile_get()
    eng.profile_enable(False)
    if one_launch is not None:
        assert (pr["sweep_launches"] == 1 and pr["update_launches"] == 0 and pr["async_fallbacks"] == 0) == one_launch, pr
    op = O.default_params(**kw)
    worst = [0.0, 0.0]
    for k in (range(B) if oracle_pairs is None else oracle_pairs):
        tgt, src, dT = host[k]
        grid = O.Grid(tgt, op)
        # voxel grid of this pair: bounds, cells, counts, f64 means, f32 inverse covariances, pca weights -- bit-exact
        mn, mx, dv, nv = eng.get_grid(k)
        omn, omx, odv = grid.bounds()
        assert np.array_equal(mn, omn) and np.array_equal(mx, omx) and np.array_equal(dv, odv)
        lv = grid.leaves()
        sel = lv[(lv["n"] >= op.min_points_per_voxel) | (lv["n"] == -1)]
        v = eng.get_voxels(k)
        assert nv == len(sel) and np.array_equal(v["idx"], sel["idx"]) and np.array_equal(v["n"], sel["n"])
        live = sel["n"] >= op.min_points_per_voxel
        assert np.array_equal(v["mean"], sel["mean"])
        assert np.array_equal(v["icov"][live], sel["icov"][live].astype(np.float32))
        if kw.get("variant", 0) == 1:
            assert np.array_equal(v["weight"][live], sel["weight"][live])
        ro = O.align(grid, src, G)
        r = res[k]
        assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"] and r["sweeps"] == ro["sweeps"], (k, r, ro)
        assert r["hits_last"] == ro["hits_last"]
        dt, dr = se3_err(ro["final"], r["final"])
        assert dt < 1e-4 and dr < 1e-5, (k, dt, dr)
        worst = [max(worst[0], dt), max(worst[1], dr)]
        assert abs(r["score"] - ro["score"]) <= 1e-9 * max(1.0, abs(ro["score"]))
        if check_motion:                      # and the registration is physically right (scene-noise level)
            dt, dr = se3_err(dT, r["final"])
            assert dt < 0.1 and dr < 0.01, (k, dt, dr)
    # the batch is deterministic run to run
    res2 = eng.batch_align(G)
    for a, b in zip(res, res2):
        assert np.array_equal(a["final"], b["final"]) and a["score"] == b["score"]
    return worst


def test_config3_shaped_batch_vs_oracle():
    """16 consecutive 65,536-pt pairs (pairs 0..15 of config 3), ndt_omp, 1 m, DIRECT7."""
    run_and_check(list(range(16)), 1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0), one_launch=False)


@pytest.mark.parametrize("mode", [ndt.DIRECT7, ndt.DIRECT1])
def test_config5_full_size_vs_oracle(mode):
    """config 5: 131,072-pt clouds, ndt_pca, 0.5 m voxels; three pairs in one device-resident batch."""
    run_and_check([0, 1, 7], 2048, dict(resolution=0.5, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=1))


def test_config3_batch_just_above_the_one_launch_threshold_vs_oracle():
    """17 pairs x 128 work items = 2,176 > 2,048 resident waves: the first batch size of config 3's shape that the engine aligns as ONE
    persistent launch by default (16 pairs, the test above, still take the rounds).  Every pair against the oracle."""
    run_and_check(list(range(16, 33)), 1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0), one_launch=True)


@pytest.mark.parametrize("mode", [ndt.DIRECT7, ndt.DIRECT1])
def test_config5_per_gpu_share_default_path_vs_oracle(mode):
    """BASELINE config 5's per-GPU share as bench.py's `other_configs` runs it: 128 pairs x 131,072 points, ndt_pca, 0.5 m -- through the
    engine's DEFAULT path, asserted to be the one-launch align (profile: one sweep launch, no update launch) --, 16 pairs spread over the
    batch against the oracle: grids bit-exact, equal iterations / hits, SE(3) inside the tolerance."""
    run_and_check(list(range(128)), 2048, dict(resolution=0.5, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=1),
                  oracle_pairs=list(range(0, 128, 8)), one_launch=True, check_motion=False)   # (parity is the point: ndt_pca at 0.5 m lands 11 cm from the simulated motion on one of the pairs, oracle and engine alike)


def test_nodelet_configuration_batch_vs_oracle():
    """what the live nodelet sets (scan_matching_odom_nodelet.cpp:109-119): ndt_pca, DIRECT1, 1 m -- 65,536-pt pairs."""
    run_and_check([0, 5, 11, 270], 1024, dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=1))


@pytest.mark.parametrize("mode,variant", [(ndt.DIRECT7, 0), (ndt.DIRECT1, 1)])
def test_results_do_not_depend_on_how_work_items_are_dealt(mode, variant, monkeypatch):
    """The sweep deals most of every per-XCD work queue statically and claims only the tail with atomics.  Which
    wave runs an item must not change a bit of the result: all-dynamic (shift 0), half, the shipped setting and almost-all-static
    (shift 6) give identical poses, scores and hit counts for a batch big enough (24 x 65,536 pts = 3,072 items) to have static
    rounds."""
    ids = list(range(100, 124))
    T, S, host, n = resident_batch(ids, 1024)
    B = len(ids)
    G = synth.default_guess()
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    ref = None
    for shift in ("0", "1", None, "6", "async"):
        if shift in (None, "async"):
            monkeypatch.delenv("MI355NDT_SWEEP_DYN_SHIFT", raising=False)
        else:
            monkeypatch.setenv("MI355NDT_SWEEP_DYN_SHIFT", shift)
        eng = ndt.Engine(ndt.default_params(**kw))          # the knob is read when the engine is created
        # the round-based kernels are what the knob deals with; the last run is the one-launch align (ndt_async.hpp), where tickets are
        # dealt to waves in yet another way: still the same bits
        eng.set_option(ndt.OPT_ASYNC_ALIGN, 1 if shift == "async" else 0)
        eng.batch_bind_device(T.data_ptr(), [n] * B, n, S.data_ptr(), [n] * B, n)
        eng.batch_build_targets()
        res = eng.batch_align(G)
        eng.close()
        if ref is None:
            ref = res
            continue
        for a, b in zip(ref, res):
            assert np.array_equal(a["final"], b["final"]) and a["score"] == b["score"] and a["iterations"] == b["iterations"]
            assert a["hits_last"] == b["hits_last"] and a["sweeps"] == b["sweeps"]


@pytest.mark.parametrize("mode,variant,res", [(ndt.DIRECT7, 0, 1.0), (ndt.DIRECT1, 1, 1.0), (ndt.DIRECT26, 0, 2.0), (ndt.KDTREE, 0, 1.0), (ndt.DIRECT7, 1, 0.5)])
def test_one_launch_align_equals_the_round_based_align(mode, variant, res):
    """MI355NDT_OPT_ASYNC_ALIGN: every pair through its own Newton loop inside ONE persistent launch (as every reference align() runs its own
    loop, ndt_omp_impl2.hpp:131-183) against the lockstep (update, sweep) rounds: identical poses, scores, iteration counts, hit counts and
    incremental transforms for a ragged batch whose pairs need different numbers of iterations -- partial rows, poses and tickets cross
    workgroups and XCDs inside the launch, so this is also the hand-off's litmus test (uneven load, every result word compared, repeated)."""
    ids = list(range(200, 240))
    T, S, host, n = resident_batch(ids, 512)
    B = len(ids)
    cnt = [n - 997 * (k % 7) for k in range(B)]             # ragged sources: the last items of a pair are short or empty
    G = np.stack([synth.default_guess() for _ in range(B)])
    G[::3, 0, 3] += 0.35                                    # some pairs start further away: iteration counts differ across the batch
    kw = dict(resolution=res, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    out = {}
    for a in (0, 1):
        eng = ndt.Engine(ndt.default_params(**kw))
        eng.set_option(ndt.OPT_ASYNC_ALIGN, a)
        assert eng.get_option(ndt.OPT_ASYNC_ALIGN) == a
        eng.batch_bind_device(T.data_ptr(), [n] * B, n, S.data_ptr(), cnt, n)
        eng.batch_build_targets()
        eng.profile_enable(True); eng.profile_reset()
        runs = [eng.batch_align(G) for _ in range(6 if a else 1)]
        pr = eng.profile_get()
        eng.profile_enable(False)
        # the option really selects the path: one launch per align and no update kernel, or rounds of (update, sweep) launches
        assert (pr["sweep_launches"] == 6 and pr["update_launches"] == 0) if a else (pr["sweep_launches"] > 3 and pr["update_launches"] > 2)
        incs = [eng.get_incremental(k) for k in (0, B // 2, B - 1)]
        eng.close()
        out[a] = (runs, incs)
    ref = out[0][0][0]
    if mode != ndt.DIRECT26:                                # (ndt_omp + DIRECT26 at 2 m oscillates: every pair runs into max_iterations + 2 = 66 iterations, the longest loop there is)
        assert len({r["iterations"] for r in ref}) >= 3     # the batch really is uneven
    for rep in out[1][0]:
        for k, (x, y) in enumerate(zip(ref, rep)):
            assert np.array_equal(x["final"], y["final"]) and x["score"] == y["score"] and x["trans_probability"] == y["trans_probability"], k
            assert x["iterations"] == y["iterations"] and x["converged"] == y["converged"] and x["sweeps"] == y["sweeps"] and x["hits_last"] == y["hits_last"], k
    for (a0, p0), (a1, p1) in zip(out[0][1], out[1][1]):
        assert np.array_equal(a0, a1) and np.array_equal(p0, p1)
    # and the oracle agrees on a sample
    op = O.default_params(**kw)
    for k in (0, 3, B - 1):
        if variant == 1 and mode in (ndt.KDTREE, ndt.DIRECT26):
            break
        t, s_, _ = host[k]
        ro = O.align(O.Grid(t, op), s_[: cnt[k]], G[k])
        assert ro["iterations"] == ref[k]["iterations"]
        dt, dr = se3_err(ro["final"], ref[k]["final"])
        assert dt < 1e-4 and dr < 1e-5


def test_one_launch_align_many_small_pairs_repeated():
    """Hand-off stress: 300 pairs of 4,096 points (8 work items per sweep, so a pair's updater changes constantly and tickets are published
    at the highest rate the engine can produce), 20 aligns in a row, every result word of every run equal to the round-based align's."""
    ids = list(range(300, 340))
    T, S, host, n = resident_batch(ids, 64)
    import torch
    reps = 8                                                  # 320 pairs: the 40 clouds, each with 8 different guesses
    Tb, Sb = T.repeat(reps, 1, 1), S.repeat(reps, 1, 1)
    B = len(ids) * reps
    G = np.stack([synth.default_guess() for _ in range(B)])
    G[:, 0, 3] += np.repeat(np.linspace(-0.3, 0.4, reps), len(ids)).astype(np.float32)
    G[:, 1, 3] += np.tile(np.linspace(-0.1, 0.1, len(ids)), reps).astype(np.float32)
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    res = {}
    for a in (0, 1):
        eng = ndt.Engine(ndt.default_params(**kw))
        eng.set_option(ndt.OPT_ASYNC_ALIGN, a)
        eng.batch_bind_device(Tb.data_ptr(), [n] * B, n, Sb.data_ptr(), [n] * B, n)
        eng.batch_build_targets()
        res[a] = [eng.batch_align(G) for _ in range(20 if a else 1)]
        eng.close()
    ref = res[0][0]
    for rep in res[1]:
        for k, (x, y) in enumerate(zip(ref, rep)):
            assert np.array_equal(x["final"], y["final"]) and x["score"] == y["score"] and x["iterations"] == y["iterations"] and x["hits_last"] == y["hits_last"], k
    torch.cuda.synchronize()


def test_host_cloud_batch_upload_paths_agree():
    """The drop-in host path for a batch: pcl::PointXYZI-style 32-byte records in ordinary host memory, staged by the engine's
    own threads (mi355ndt_batch_set_clouds), by several caller threads (batch_set_target / batch_set_source on different pairs)
    and one cloud at a time -- ragged cloud sizes -- all give the bits of the device-resident run of the same pairs
    (the call pattern being batched: scan_matching_odom_nodelet.cpp:220-221)."""
    import threading
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    pairs = []
    for k in range(6):
        t, s, dT = synth.make_pair(40 + k, 256 if k % 2 else 192)
        t, s = t.numpy(), s.numpy()[: len(s) - 1000 * k - 1]
        t32 = np.zeros((len(t), 8), np.float32); t32[:, :3] = t; t32[:, 3] = 1.0; t32[:, 4] = 7.0
        s32 = np.zeros((len(s), 8), np.float32); s32[:, :3] = s; s32[:, 3] = 1.0; s32[:, 4] = 9.0
        pairs.append((t32, s32))
    B = len(pairs)
    G = synth.default_guess()
    mt, ms = max(len(p[0]) for p in pairs), max(len(p[1]) for p in pairs)

    def run(fill):
        eng = ndt.Engine(ndt.default_params(**kw))
        eng.batch_reserve(B, mt, ms)
        fill(eng)
        eng.batch_build_targets()
        r = eng.batch_align(G)
        v = eng.get_voxels(B - 1)
        eng.close()
        return r, v

    def one_by_one(eng):
        for k, (t, s) in enumerate(pairs):
            eng.batch_set_target(k, t)
            eng.batch_set_source(k, s)

    def own_threads(eng):
        tp = np.array([p[0].ctypes.data for p in pairs], np.uint64); tc = np.array([len(p[0]) for p in pairs], np.uint64)
        sp = np.array([p[1].ctypes.data for p in pairs], np.uint64); sc = np.array([len(p[1]) for p in pairs], np.uint64)
        eng.batch_set_clouds_raw(0, tp, tc, sp, sc, 32, threads=4)

    def caller_threads(eng):
        def up(w):
            for k in range(w, B, 3):
                eng.batch_set_source_raw(k, pairs[k][1].ctypes.data, len(pairs[k][1]), 32)
                eng.batch_set_target_raw(k, pairs[k][0].ctypes.data, len(pairs[k][0]), 32)
        th = [threading.Thread(target=up, args=(w,)) for w in range(3)]
        [t.start() for t in th]; [t.join() for t in th]

    ref, vref = run(one_by_one)
    for fill in (own_threads, caller_threads):
        got, v = run(fill)
        for a, b in zip(ref, got):
            assert np.array_equal(a["final"], b["final"]) and a["score"] == b["score"] and a["iterations"] == b["iterations"]
        assert np.array_equal(v["mean"], vref["mean"]) and np.array_equal(v["icov"], vref["icov"])
    # and against the oracle
    for k in (0, B - 1):
        ro = O.align(O.Grid(pairs[k][0][:, :3].copy(), O.default_params(**kw)), pairs[k][1][:, :3].copy(), G)
        assert ref[k]["iterations"] == ro["iterations"]
        dt, dr = se3_err(ro["final"], ref[k]["final"])
        assert dt < 1e-4 and dr < 1e-5
