"""Latency mode: the scan-matching node's per-frame loop -- ScanMatchingOdomNodelet::matching_s2k,
src/lidar_odometry/scan_matching_odom_nodelet.cpp:192-261 -- tracked on the device by mi355ndt_sequence_run, and the opt-in
fine-grained sweep behind it.

CPU: the oracle's policy step (ora_policy_step, the arithmetic of :229-250 with an explicit operation order) against NumPy, and the
oracle's whole-run driver (oracle_py.sequence) against the host-side Python policy (lv_slam_amd/odometry.py) on a scripted
registration.  GPU: the device run against the oracle run -- same keyframes, same iteration counts, poses inside the north-star
tolerance -- at test size and at BASELINE size (65,536 points, the nodelet's own configuration, 65 frames)."""
import ctypes as C

import numpy as np
import pytest

from conftest import se3_err
from lv_slam_amd.odometry import ScanMatchingOdometry
from oracle import oracle_py as O


def _rand_rigid(rng, t_scale=2.0, a_scale=0.2):
    w = rng.normal(size=3) * a_scale
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = rng.normal(size=3) * t_scale
    return M


def test_policy_step_matches_numpy():
    """ora_policy_step (:229-250) == the same formulas through numpy.linalg, to f64 rounding; keyframe rule and state updates."""
    rng = np.random.default_rng(5)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for trial in range(50):
        pre, key_pose = _rand_rigid(rng), _rand_rigid(rng, 30.0)
        fin = _rand_rigid(rng, 3.0).astype(np.float32)
        stamp0, stamp = 10.0, 10.0 + rng.uniform(0, 2)
        thr = np.array([rng.uniform(1, 8), 0.17, 1.0])
        pre_c, kp_c, ks = pre.ravel().copy(), key_pose.ravel().copy(), C.c_double(stamp0)
        fin_cm = np.ascontiguousarray(fin.T).ravel().copy()
        odom, guess, test = np.zeros(16), np.zeros(16), np.zeros(3)
        key = O.lib().ora_policy_step(p(pre_c), p(kp_c), C.byref(ks), p(fin_cm), stamp, p(thr), p(odom), p(guess), p(test))
        tf = fin.astype(np.float64)
        s2s = np.linalg.inv(pre) @ tf
        want_odom = key_pose @ tf
        dx = np.linalg.norm(tf[:3, 3])
        from lv_slam_amd.odometry import quaternionf_w
        da = 2.0 * np.arccos(np.float64(quaternionf_w(tf[:3, :3])))
        want_key = dx > thr[0] or da > thr[1] or (stamp - stamp0) > thr[2]
        assert bool(key) == bool(want_key)
        assert np.allclose(odom.reshape(4, 4), want_odom, rtol=0, atol=1e-12)
        assert abs(test[0] - dx) < 1e-12 and abs(test[1] - da) < 1e-6 and test[2] == stamp - stamp0     # da goes through acosf
        new_pre = np.eye(4) if want_key else tf
        assert np.allclose(pre_c.reshape(4, 4), new_pre, atol=1e-15)
        assert np.allclose(guess.reshape(4, 4), new_pre @ s2s, rtol=0, atol=1e-11)
        assert np.allclose(kp_c.reshape(4, 4), want_odom if want_key else key_pose, atol=1e-12)
        assert ks.value == (stamp if want_key else stamp0)


def test_oracle_sequence_follows_the_host_policy(monkeypatch):
    """oracle_py.sequence (the checker of the device run) takes the same decisions as lv_slam_amd/odometry.py on a scripted
    registration: frame-1 double align, keyframe switches, guess propagation, odometry across the switches."""
    finals = {}

    def fake_align(grid, src, guess):
        k, key = int(src[0, 0]), grid
        F = np.eye(4, dtype=np.float32)
        F[0, 3] = np.float32(1.2 * (k - key))
        F[1, 3] = np.float32(0.01 * k)
        finals.setdefault(k, []).append(np.array(guess, np.float64))
        return dict(final=F, iterations=3, converged=True, trans_probability=1.0)

    monkeypatch.setattr(O, "align", fake_align)
    monkeypatch.setattr(O, "Grid", lambda pts, prm: int(pts[0, 0]))
    frames = [np.full((4, 3), float(k), np.float32) for k in range(9)]
    stamps = [0.1 * k for k in range(9)]
    seq = O.sequence(frames, stamps, None, keyframe_delta_trans=3.0, keyframe_delta_angle=0.17, keyframe_delta_time=1e9)

    class Reg:
        def setInputTarget(self, c): self.key = int(c[0, 0])
        def setInputSource(self, c): self.src = c
        def align(self, g): self.r = fake_align(self.key, self.src, g)
        def getFinalTransformation(self): return self.r["final"]

    od = ScanMatchingOdometry(Reg(), keyframe_delta_trans=3.0, keyframe_delta_angle=0.17, keyframe_delta_time=1e9)
    keys = []
    for k in range(9):
        pose, _ = od.cloud_callback(stamps[k], frames[k])
        assert np.allclose(seq[k]["odom"], pose, atol=1e-9), k
        keys.append(od.key_id)
    assert [f["key_id"] for f in seq] == [0, 0, 0, 0, 3, 3, 3, 6, 6]          # the keyframe each scan was MATCHED against
    assert [k for k, f in enumerate(seq) if f["new_keyframe"]] == [0, 3, 6] and keys[-1] == 6
    assert seq[1]["aligns"] == 2 and all(f["aligns"] == 1 for f in seq[2:])
    assert len(finals[1]) == 4                                                # two aligns of frame 1, by both drivers


# ------------------------------------------------------------------------------------------------------------------ GPU
def _nodelet_params(ndt_mod, **kw):
    base = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=3, variant=1)      # pclpca, DIRECT1 (:109-119)
    base.update(kw)
    return ndt_mod.default_params(**base)


def _compare_runs(dev, ora, tol_t=1e-4, tol_r=1e-5):
    assert len(dev) == len(ora)
    for k, (d, o) in enumerate(zip(dev, ora)):
        assert d["key_id"] == o["key_id"] and d["new_keyframe"] == o["new_keyframe"], k
        if k == 0:
            continue
        assert d["iterations"] == o["iterations"] and d["converged"] == o["converged"] and d["aligns"] == o["aligns"], (k, d["iterations"], o["iterations"])
        dt, dr = se3_err(o["tf_s2k"], d["tf_s2k"])
        assert dt < tol_t and dr < tol_r, (k, dt, dr)
        dt, dr = se3_err(o["odom"], d["odom"])
        assert dt < tol_t and dr < tol_r, (k, dt, dr)
        assert abs(d["test"][0] - o["test"][0]) < 1e-4 and abs(d["test"][2] - o["test"][2]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode,variant", [(3, 1), (2, 0)])
def test_device_sequence_vs_oracle_small(mode, variant):
    """12 frames of 8,192 points: the device run (guess propagation, keyframe test, frame-1 double align and target switch decided on
    the device) against the oracle run; several keyframe switches (2.5 m rule, and the 1 s rule at 10 Hz)."""
    from lv_slam_amd import ndt, synth
    scans, truth = synth.make_sequence(12, 256, n_beams=32)
    scans = [s.numpy() for s in scans]
    stamps = [0.1 * k for k in range(len(scans))]
    eng = ndt.Engine(_nodelet_params(ndt, neighbor_mode=mode, variant=variant))
    dev, stats = eng.sequence_run(scans, stamps, keyframe_delta_trans=2.5)
    ora = O.sequence(scans, stamps, O.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant),
                     keyframe_delta_trans=2.5)
    _compare_runs(dev, ora)
    assert sum(f["new_keyframe"] for f in dev) >= 3 and stats["aligns"] == len(scans) and stats["track_ms"] > 0
    # the run is deterministic, and the engine is usable as an ordinary registration object afterwards
    dev2, _ = eng.sequence_run(scans, stamps, keyframe_delta_trans=2.5)
    for a, b in zip(dev, dev2):
        assert np.array_equal(a["odom"], b["odom"]) and np.array_equal(a["tf_s2k"], b["tf_s2k"])
    eng.set_target(scans[0])
    eng.set_source(scans[1])
    G = np.eye(4, dtype=np.float32)
    G[0, 3] = 1.5
    r = eng.align(G)
    prm = O.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    ro = O.align(O.Grid(scans[0], prm), scans[1], G)
    assert r["iterations"] == ro["iterations"]
    # against the drive itself: sparse 8,192-point scans, a dozen frames of drift -- a sanity bound, not an accuracy claim
    # (device and oracle agree with each other to 1e-4 m above; this is how far BOTH are from the true motion)
    dt, dr = se3_err(np.linalg.inv(truth[0]) @ truth[-1], dev[-1]["odom"])
    assert dt < 2.5 and dr < 0.2, (dt, dr)


@pytest.mark.gpu
def test_device_sequence_vs_oracle_full_size():
    """BASELINE size: 65 frames of 65,536 points, the nodelet's configuration (pclpca, 1.0 m, DIRECT1, eps 0.01, 64 iterations,
    scan_matching_odom_nodelet.cpp:109-119) and its keyframe thresholds (:67-76: 5 m, 0.17 rad, 1 s at 10 Hz): trajectory == oracle
    trajectory, frame by frame."""
    import torch
    from lv_slam_amd import ndt, synth
    scans, truth = synth.make_sequence(65, 1024, device="cuda")
    scans = [s.cpu().numpy() for s in scans]
    stamps = [0.1 * k for k in range(len(scans))]
    eng = ndt.Engine(_nodelet_params(ndt))
    dev, stats = eng.sequence_run(scans, stamps)
    ora = O.sequence(scans, stamps, O.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=3, variant=1))
    _compare_runs(dev, ora)
    assert sum(f["new_keyframe"] for f in dev) >= 6
    per_frame_ms = stats["track_ms"] / (len(scans) - 1)
    print(f"sequence: {len(scans)} frames, track {stats['track_ms']:.2f} ms = {per_frame_ms:.3f} ms per frame, build {stats['build_ms']:.2f} ms, "
          f"upload {stats['upload_ms']:.1f} ms, update launches {stats['update_launches']}")
    assert per_frame_ms < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode,variant,tiles", [(3, 1, "2"), (3, 1, "1"), (2, 0, "2"), (2, 1, "2"), (3, 0, "1")])
def test_latency_mode_single_registration_vs_oracle(mode, variant, tiles, monkeypatch):
    """The opt-in fine-grained sweep (work items of 128 / 64 points, its own fixed f64 tree): one sweep inside the 1e-11 bar, align
    with the oracle's iteration count and pose, bit-identical from run to run, and no effect on an engine that did not opt in."""
    from lv_slam_amd import ndt, synth
    monkeypatch.setenv("MI355NDT_FINE_TILES", tiles)
    tgt, src, _ = synth.make_pair(3, 1024)
    tgt, src = tgt.numpy(), src.numpy()
    src = src[:65000]                                      # a ragged tail: the last 4-row chunk is partly empty
    prm_kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    eng = ndt.Engine(ndt.default_params(**prm_kw))
    eng.set_latency_mode(True)
    eng.set_target(tgt)
    eng.set_source(src)
    grid = O.Grid(tgt, O.default_params(**prm_kw))
    p = np.array([0.9, 0.02, -0.01, 0.003, -0.002, 0.01])
    s, g, H, hits = eng.derivatives(p)
    so, go, Ho, hits_o = O.derivatives_at(grid, src, p)
    scale = max(np.abs(Ho).max(), np.abs(go).max(), abs(so))
    assert hits == hits_o and abs(s - so) < 1e-11 * scale and np.abs(g - go).max() < 1e-11 * scale and np.abs(H - Ho).max() < 1e-11 * scale
    G = synth.default_guess()
    r = eng.align(G)
    ro = O.align(grid, src, G)
    assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"]
    dt, dr = se3_err(ro["final"], r["final"])
    assert dt < 1e-4 and dr < 1e-5
    r2 = eng.align(G)
    assert np.array_equal(r["final"], r2["final"]) and r["score"] == r2["score"]
    plain = ndt.Engine(ndt.default_params(**prm_kw))
    plain.set_target(tgt)
    plain.set_source(src)
    rp = plain.align(G)
    assert rp["iterations"] == r["iterations"]
    dt, dr = se3_err(rp["final"], r["final"])
    assert dt < 1e-4 and dr < 1e-5


@pytest.mark.gpu
def test_sequence_run_edge_cases():
    """ragged frames (every frame its own point count), runs of one and two frames, what the entry point refuses, and the engine's
    ordinary surface after a run."""
    from lv_slam_amd import ndt, synth
    scans, _ = synth.make_sequence(7, 256, n_beams=32)
    scans = [s.numpy() for s in scans]
    ragged = [s[: len(s) - 137 * k] for k, s in enumerate(scans)]            # 8192, 8055, 7918, ... points
    stamps = [0.1 * k for k in range(len(ragged))]
    prm_kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=3, variant=1)
    eng = ndt.Engine(ndt.default_params(**prm_kw))
    dev, stats = eng.sequence_run(ragged, stamps, keyframe_delta_trans=2.0)
    ora = O.sequence(ragged, stamps, O.default_params(**prm_kw), keyframe_delta_trans=2.0)
    _compare_runs(dev, ora)
    one, st1 = eng.sequence_run(ragged[:1], stamps[:1])
    assert len(one) == 1 and one[0]["new_keyframe"] and np.array_equal(one[0]["odom"], np.eye(4)) and st1["aligns"] == 0
    two, st2 = eng.sequence_run(ragged[:2], stamps[:2])
    assert two[1]["aligns"] == 2 and st2["aligns"] == 2
    dt, dr = se3_err(ora[1]["tf_s2k"], two[1]["tf_s2k"])
    assert two[1]["iterations"] == ora[1]["iterations"] and dt < 1e-4 and dr < 1e-5
    # refused: an empty frame; KDTREE and the live More-Thuente configuration are served by the batch kernels only
    with pytest.raises(ndt.NDTError) as e:
        eng.sequence_run([ragged[0], np.zeros((0, 3), np.float32)], [0.0, 0.1])
    assert e.value.code == -2
    for kw in (dict(neighbor_mode=0), dict(step_size=0.004)):
        bad = ndt.Engine(ndt.default_params(**{**prm_kw, **kw}))
        with pytest.raises(ndt.NDTError) as e:
            bad.sequence_run(ragged[:3], stamps[:3])
        assert e.value.code == -6
    # the handle is an ordinary registration object again afterwards (batch mode unless asked otherwise)
    eng.set_target(scans[0])
    eng.set_source(scans[1])
    G = np.eye(4, dtype=np.float32)
    G[0, 3] = 1.5
    r = eng.align(G)
    ro = O.align(O.Grid(scans[0], O.default_params(**prm_kw)), scans[1], G)
    assert r["iterations"] == ro["iterations"] and np.array_equal(r["final"], ro["final"])


@pytest.mark.gpu
def test_latency_mode_small_batch_vs_oracle():
    """Latency mode is chosen for any SMALL batch, not only for one pair: five pairs of different sizes through the fine-grained sweep,
    the block-level chunk sums and the pump, each against the oracle and against the batch-mode engine."""
    from lv_slam_amd import ndt, synth
    prm_kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=2, variant=0)
    pairs = []
    for k, cut in enumerate((0, 1000, 4097, 130, 7777)):
        t, s, _ = synth.make_pair(20 + k, 512)
        pairs.append((t.numpy(), s.numpy()[: 32768 - cut]))
    G = synth.default_guess()
    out = {}
    for lat in (False, True):
        eng = ndt.Engine(ndt.default_params(**prm_kw))
        eng.set_latency_mode(lat)
        eng.batch_reserve(len(pairs), 32768, 32768)
        for k, (t, s) in enumerate(pairs):
            eng.batch_set_target(k, t)
            eng.batch_set_source(k, s)
        out[lat] = eng.batch_align(np.broadcast_to(G, (len(pairs), 4, 4)))
        again = eng.batch_align(np.broadcast_to(G, (len(pairs), 4, 4)))
        for a, b in zip(out[lat], again):
            assert np.array_equal(a["final"], b["final"]) and a["score"] == b["score"]
    op = O.default_params(**prm_kw)
    for k, (t, s) in enumerate(pairs):
        ro = O.align(O.Grid(t, op), s, G)
        for lat in (False, True):
            r = out[lat][k]
            assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"], (k, lat)
            dt, dr = se3_err(ro["final"], r["final"])
            assert dt < 1e-4 and dr < 1e-5, (k, lat, dt, dr)
