#!/usr/bin/env python3
"""Randomised differential test: the HIP path (through the C-ABI) against the oracle on seeded random cases -- GPU box only.

Every case draws a scene (synthetic scan pair or random planes / blobs, optionally far from the origin, with non-finite and
duplicated points), a parameter set (variant, neighbour mode, resolution, step size, outlier ratio, epsilon, iteration cap,
min_points_per_voxel, eigenvalue factor), a guess near the true motion, and an entry path (single-pair API, a ragged batch,
latency mode), then checks what tests/test_gpu_parity.py checks on its fixed cases: voxel grid exact, sweep within rtol 1e-11,
align with the same iteration count / convergence flag / sweep count and the pose inside (1e-4 m, 1e-5 rad).

  python tools/fuzz_parity.py --cases 300 --seed 1            -> gpurun_out/fuzz/fuzz_<seed>.json (+ one line per failure on stdout)

Test infrastructure (it calls the oracle); nothing in the product imports it."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from lv_slam_amd import ndt, synth          # noqa: E402
from oracle import oracle_py as O           # noqa: E402
from conftest import se3_err                # noqa: E402

MODES = [ndt.DIRECT7, ndt.DIRECT1, ndt.DIRECT26, ndt.KDTREE]


def small_motion(rng, scale=1.0):
    p = np.concatenate([rng.normal(0, 0.15 * scale, 3), rng.normal(0, 0.01 * scale, 3)])
    return O.se3_exp(p)


def scene(rng, case):
    """-> (target [n,3] f32, source [m,3] f32, guess 4x4 f32)"""
    kind = rng.choice(["scan", "scan", "planes", "blobs"])
    if kind == "scan":
        az = int(rng.choice([64, 128, 256, 512, 1024], p=[0.25, 0.25, 0.25, 0.2, 0.05]))
        beams = int(rng.choice([16, 32, 64]))
        tgt, src, dT = synth.make_pair(int(rng.integers(0, 4000)), az, n_beams=beams)
        tgt, src = tgt.numpy().copy(), src.numpy().copy()
        G = synth.default_guess().astype(np.float64)
    else:
        n = int(rng.integers(200, 6000))
        if kind == "planes":
            pts = []
            for _ in range(int(rng.integers(2, 7))):
                o, u, v = rng.uniform(-15, 15, 3), rng.normal(size=3), rng.normal(size=3)
                u /= np.linalg.norm(u); v -= u * (u @ v); v /= np.linalg.norm(v)
                k = n // 4 + 1
                pts.append(o + np.outer(rng.uniform(-8, 8, k), u) + np.outer(rng.uniform(-8, 8, k), v) + rng.normal(0, 0.03, (k, 3)))
            tgt = np.concatenate(pts)
        else:
            c = rng.uniform(-20, 20, (int(rng.integers(3, 40)), 3))
            tgt = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.05, 1.0), (n, 3))
        M = small_motion(rng)
        sel = rng.permutation(len(tgt))[: max(1, int(len(tgt) * rng.uniform(0.3, 1.0)))]
        src = (tgt[sel] - M[:3, 3]) @ M[:3, :3] + rng.normal(0, 0.01, (len(sel), 3))       # src = M^-1 tgt
        G = M @ small_motion(rng, 0.3)
    tgt, src = tgt.astype(np.float32), src.astype(np.float32)
    if rng.random() < 0.25:                       # far from the origin (cell indices / f32 keys away from zero), both clouds + guess consistent
        off = rng.choice([-1.0, 1.0], 3) * rng.choice([50.0, 300.0, 2000.0]) * rng.uniform(0.5, 1.0, 3)
        tgt = (tgt + off).astype(np.float32)
        Gt = np.eye(4); Gt[:3, 3] = off
        G = Gt @ G
    if rng.random() < 0.2:                        # non-finite points on either side
        tgt[rng.integers(0, len(tgt), max(1, len(tgt) // 50)), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf])
        src[rng.integers(0, len(src), max(1, len(src) // 50)), rng.integers(0, 3)] = np.nan
    if rng.random() < 0.2:                        # duplicated points (equal keys, equal coordinates)
        tgt = np.concatenate([tgt, tgt[rng.integers(0, len(tgt), len(tgt) // 5)]])
    if rng.random() < 0.05:                       # a stray point far outside everything (the guards of the cell arithmetic)
        which = tgt if rng.random() < 0.6 else src
        which[int(rng.integers(0, len(which)))] = rng.choice([-1.0, 1.0], 3) * float(rng.choice([1e4, 1e7, 1e12, 1e30]))
    if rng.random() < 0.15:                       # a ragged, tiny source
        src = src[: int(rng.integers(1, 130))]
    return tgt, src, G.astype(np.float32)


def params(rng):
    kw = dict(variant=int(rng.integers(0, 2)), neighbor_mode=int(rng.choice(MODES, p=[0.4, 0.3, 0.15, 0.15])),
              resolution=float(rng.choice([0.5, 0.7, 1.0, 1.0, 1.3, 2.0, 3.7])),
              trans_epsilon=float(rng.choice([0.01, 0.01, 0.001, 0.05])), max_iterations=int(rng.choice([1, 2, 5, 35, 64])),
              outlier_ratio=float(rng.choice([0.55, 0.55, 0.3, 0.8])), step_size=float(rng.choice([0.1, 0.1, 0.05, 0.5])),
              min_points_per_voxel=int(rng.choice([6, 6, 3, 10])), min_covar_eigvalue_mult=float(rng.choice([0.01, 0.01, 0.1])))
    if rng.random() < 0.08:                       # the live More-Thuente configuration (step_size <= eps / 2)
        kw["step_size"], kw["trans_epsilon"], kw["max_iterations"] = 0.004, 0.01, int(rng.choice([3, 8]))
    return kw


AUX_P = 0.15          # share of cases that also check the surfaces around align (--aux)


ORA_VARIANTS = [("acc_chunk_8", 0, 8), ("acc_chunk_2048", 0, 2048), ("solve_lu", 16, 256), ("solve_svd_two_sided", 32, 256)]


def classify(align_again, r_gpu, r_canonical):
    """A failing align is replayed with the oracle's own legitimate variants (tools/order_sensitivity.py: another f64 partial-sum
    length, LU / two-sided SVD for the Newton solve).  If one of them moves the ORACLE away from its canonical result by more than
    the tolerance -- or lands on the HIP result -- the case is order-sensitive (a chaotic run), not a defect of the HIP path."""
    import ctypes as C
    L = O.lib()
    L.ora_set_variant.argtypes = [C.c_uint, C.c_int]
    out = {}
    try:
        for name, flags, chunk in ORA_VARIANTS:
            L.ora_set_variant(flags, chunk)
            rv = align_again()
            dt, dr = se3_err(r_canonical["final"], rv["final"])
            gt, gr = se3_err(r_gpu["final"], rv["final"])
            out[name] = dict(it=int(rv["iterations"]), d_vs_canonical=[float(dt), float(dr)], d_vs_hip=[float(gt), float(gr)])
    finally:
        L.ora_set_variant(0, 256)
    sens = any(v["d_vs_canonical"][0] > 1e-4 or v["d_vs_canonical"][1] > 1e-5 or v["it"] != r_canonical["iterations"] for v in out.values())
    same = [k for k, v in out.items() if v["d_vs_hip"][0] < 1e-9 and v["d_vs_hip"][1] < 1e-9 and v["it"] == r_gpu["iterations"]]
    return dict(order_sensitive=bool(sens), hip_equals_oracle_variant=same, variants=out)


def ora_align(grid, src, G):
    """O.align, plus the case the oracle's wrapper refuses: no grid at all (the leaf-too-small guard, voxel_grid_covariance_omp_impl.hpp:
    75-84) is an empty grid -- nothing is hit, the first Newton step is zero: converged, 0 iterations, final == guess (impl2:147-152)."""
    if not grid.ok:
        return dict(iterations=0, converged=True, sweeps=1, final=np.asarray(G, np.float32).copy(), hits_last=0, score=0.0, trans_probability=0.0,
                    transformation=np.eye(4, dtype=np.float32), previous_transformation=np.eye(4, dtype=np.float32), no_grid=True)
    return O.align(grid, src, G)


ARITH = 0
STATS = {"aligns": 0, "iterations": 0, "zero_hit_aligns": 0, "not_converged": 0, "hit_iteration_cap": 0, "paths": {}, "modes": {}, "mt_live": 0,
         "voxel_checks": 0, "sweep_checks": 0, "searchable_leaves": 0}


def compare(r, ro, what, fails, ctx, align_again=None):
    STATS["aligns"] += 1
    STATS["iterations"] += int(ro["iterations"])
    STATS["zero_hit_aligns"] += int(ro["hits_last"] == 0)
    STATS["not_converged"] += int(not ro["converged"])
    STATS["hit_iteration_cap"] += int(ro["iterations"] >= ctx["max_iterations"])
    ok = r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"] and r["sweeps"] == ro["sweeps"]
    dt, dr = se3_err(ro["final"], r["final"])
    if not (np.isfinite(dt) and np.isfinite(dr)):
        same = np.array_equal(np.asarray(r["final"]), np.asarray(ro["final"]), equal_nan=True)
        ok = ok and same
    else:
        ok = ok and dt < 1e-4 and dr < 1e-5
    if not ok and ARITH and r.get("status") == 1:      # MI355NDT_WARN_TOLERANCE_ARITH: the engine itself says the tolerance arithmetic is not to be trusted here
        STATS["warned_and_outside"] = STATS.get("warned_and_outside", 0) + 1
        return ok
    if ARITH and r.get("status") == 1:
        STATS["warned_and_inside"] = STATS.get("warned_and_inside", 0) + 1
    if not ok:
        f = dict(ctx, what=what, it=[r["iterations"], ro["iterations"]], conv=[bool(r["converged"]), bool(ro["converged"])],
                 sweeps=[r["sweeps"], ro["sweeps"]], dtrans=float(dt), drot=float(dr), hits_last=int(ro["hits_last"]))
        if align_again is not None:
            f["classification"] = classify(align_again, r, ro)
        fails.append(f)
    return ok


def run_aux(case, rng, kw, fails, oracle_only):
    """The surfaces around align: the device prefilter (prefiltering_nodelet.cpp:137-181) against ora_prefilter -- same points, same
    f32 centroids, same order; getFitnessScore(max_range) (loop_detector.hpp:249-262); the output cloud of align and
    getLastIncrementalTransformation's two matrices against the oracle's."""
    ctx = dict(case=case, path="aux", **kw)
    STATS["paths"]["aux"] = STATS["paths"].get("aux", 0) + 1
    tgt, src, G = scene(rng, case)
    raw = np.concatenate([tgt, src]) if rng.random() < 0.5 else tgt
    if rng.random() < 0.5:
        raw = raw - raw[np.isfinite(raw).all(1)].mean(0).astype(np.float32)          # around the sensor, where the distance gate bites
    near, far = float(rng.choice([0.5, 1.0, 3.0])), float(rng.choice([100.0, 40.0, 8.0]))
    leaf = float(rng.choice([0.1, 0.0, 0.25, 1.0, 1e-4, 0.033]))
    use_gate = bool(rng.random() < 0.8)
    exp = O.prefilter(raw, near, far, leaf, use_gate)
    op = O.default_params(**kw)
    if oracle_only:
        return
    eng = ndt.Engine(ndt.default_params(**kw))
    try:
        got = eng.prefilter(raw, near, far, leaf, use_gate)
        STATS["prefilter_checks"] = STATS.get("prefilter_checks", 0) + 1
        if got.shape != exp.shape or not np.array_equal(got, exp, equal_nan=True):
            fails.append(dict(ctx, what="prefilter", near=near, far=far, leaf=leaf, gate=use_gate, shapes=[list(got.shape), list(exp.shape)]))
        fin_t = tgt[np.isfinite(tgt).all(1)]
        if len(fin_t) == 0:
            return
        eng.set_target(tgt); eng.set_source(src)
        r = eng.align(G)
        ro = ora_align(O.Grid(tgt, op), src, G)
        if r["iterations"] == ro["iterations"] and np.array_equal(np.asarray(r["final"]), np.asarray(ro["final"])) and not (ARITH and eng.get_option(7)):   # (bit comparisons: default arithmetic only)
            # (only meaningful when the two aligns ended on the same bits: see the chaotic ndt_pca / DIRECT26 runs)
            STATS["aux_align_checks"] = STATS.get("aux_align_checks", 0) + 1
            a, b = eng.get_incremental()
            if not (np.array_equal(a, ro["transformation"], equal_nan=True) and np.array_equal(b, ro["previous_transformation"], equal_nan=True)):
                # exp(delta_p) of the last two steps in f32: the oracle's own Newton-solve variants (LU, two-sided SVD) move delta_p by
                # ~1e-8 relative when H is ill-conditioned, which is a last f32 bit in a few entries -- replay them before calling it a defect
                import ctypes as C
                L = O.lib()
                L.ora_set_variant.argtypes = [C.c_uint, C.c_int]
                var, sens, same = {}, False, []
                try:
                    for name, flags, chunk in ORA_VARIANTS:
                        L.ora_set_variant(flags, chunk)
                        rv = ora_align(O.Grid(tgt, op), src, G)
                        ta, tb = np.asarray(rv["transformation"]), np.asarray(rv["previous_transformation"])
                        d_can = not (np.array_equal(ta, ro["transformation"], equal_nan=True) and np.array_equal(tb, ro["previous_transformation"], equal_nan=True))
                        d_hip = not (np.array_equal(ta, a, equal_nan=True) and np.array_equal(tb, b, equal_nan=True))
                        var[name] = dict(differs_from_canonical=bool(d_can), differs_from_hip=bool(d_hip))
                        sens = sens or d_can
                        if not d_hip:
                            same.append(name)
                finally:
                    L.ora_set_variant(0, 256)
                ulp = float(np.nanmax(np.abs(a.astype(np.float64) - np.asarray(ro["transformation"], np.float64)) / np.maximum(np.spacing(np.abs(np.asarray(ro["transformation"], np.float32))).astype(np.float64), 1e-45)))
                fails.append(dict(ctx, what="incremental transforms", max_f32_ulps=ulp,
                                  classification=dict(order_sensitive=bool(sens), hip_equals_oracle_variant=same, variants=var)))
            out = eng.get_aligned()
            F = np.asarray(ro["final"], np.float32)
            s32 = src.astype(np.float32)
            want = np.stack([((F[a_, 0] * s32[:, 0] + F[a_, 1] * s32[:, 1]) + F[a_, 2] * s32[:, 2]) + F[a_, 3] for a_ in range(3)], 1).astype(np.float32)
            if not np.array_equal(out, want, equal_nan=True):
                fails.append(dict(ctx, what="aligned cloud", max_abs=float(np.nanmax(np.abs(out - want)))))
            for mr in (float(rng.choice([0.04, 1.0, 25.0])), float("inf")):
                gs, gn = eng.fitness_score(mr)
                es, en = O.fitness_score(tgt, src, F, mr)
                ok = gn == en and (abs(gs - es) <= 1e-12 * max(1.0, abs(es)) or (gs == es))
                if not ok:
                    fails.append(dict(ctx, what="fitness score", max_range=mr, got=[float(gs), int(gn)], want=[float(es), int(en)]))
    finally:
        eng.close()


def run_sequence(case, rng, kw, fails, oracle_only):
    """mi355ndt_sequence_run against the oracle driving the same call-site policy (oracle_py.sequence)."""
    kw = dict(kw)
    kw["neighbor_mode"] = int(rng.choice([ndt.DIRECT7, ndt.DIRECT1]))
    if kw["step_size"] <= kw["trans_epsilon"] / 2:
        kw["step_size"] = 0.1
    gp, op = ndt.default_params(**kw), O.default_params(**kw)
    n = int(rng.integers(2, 9))
    az, beams = int(rng.choice([64, 128, 256])), int(rng.choice([16, 32, 64]))
    frames, _ = synth.make_sequence(n, az, n_beams=beams, seed=int(rng.integers(1, 1 << 30)))
    frames = [f.numpy() for f in frames]
    stamps = np.cumsum(rng.uniform(0.05, 0.6, n))
    thr = (float(rng.choice([0.5, 2.0, 5.0])), float(rng.choice([0.02, 0.17])), float(rng.choice([0.3, 1.0, 100.0])))
    ctx = dict(case=case, path="sequence", frames=n, thresholds=thr, **kw)
    STATS["paths"]["sequence"] = STATS["paths"].get("sequence", 0) + 1
    ref = O.sequence(frames, stamps, op, *thr)
    if oracle_only:
        return
    eng = ndt.Engine(gp)
    try:
        got, _ = eng.sequence_run(frames, stamps, *thr)
        for k in range(n):
            a, b = got[k], ref[k]
            STATS["aligns"] += int(k > 0)
            STATS["iterations"] += int(b["iterations"])
            dt, dr = se3_err(b["odom"], a["odom"])
            ok = (a["key_id"] == b["key_id"] and a["new_keyframe"] == b["new_keyframe"] and a["iterations"] == b["iterations"]
                  and a["converged"] == b["converged"] and dt < 1e-4 * max(1, k) and dr < 1e-5 * max(1, k))
            if not ok:
                fails.append(dict(ctx, what=f"sequence frame {k}", key=[a["key_id"], b["key_id"]], newkey=[a["new_keyframe"], b["new_keyframe"]],
                                  it=[a["iterations"], b["iterations"]], dtrans=float(dt), drot=float(dr)))
                break
    finally:
        eng.close()


STREAM_P = 0.0   # --stream: probability that a case goes through mi355ndt_stream_* instead of one of the other entry paths


def run_stream(case, rng, kw, fails, oracle_only):
    """Three ragged batches of 2..5 pairs through mi355ndt_stream_begin / _submit / _collect (2..4 contexts, a random hand-over threshold,
    device-resident SoA clouds) against the oracle pair by pair.  Non-finite points are dropped before the upload on both sides: the stream
    binds device buffers, which carry finite points (the host-cloud entry points do that filtering themselves)."""
    import torch
    ctx = dict(case=case, path="stream", **kw)
    STATS["paths"]["stream"] = STATS["paths"].get("stream", 0) + 1
    gp, op = ndt.default_params(**kw), O.default_params(**kw)
    n_pairs = int(rng.integers(2, 6))
    nctx = int(rng.integers(2, 5))
    thresh = int(rng.choice([-1, 0, 1, 3]))
    batches = []
    for _ in range(3):
        sc = []
        for _ in range(n_pairs):
            t, s_, G = scene(rng, case)
            t = np.ascontiguousarray(t[np.isfinite(t).all(1)], np.float32)
            s_ = np.ascontiguousarray(s_[np.isfinite(s_).all(1)], np.float32)
            if len(t) == 0 or len(s_) == 0:
                t, s_, G = scene(np.random.default_rng([case, 7]), 1)        # (a plain scan pair)
                t = np.ascontiguousarray(t[np.isfinite(t).all(1)], np.float32); s_ = np.ascontiguousarray(s_[np.isfinite(s_).all(1)], np.float32)
            sc.append((t, s_, G))
        batches.append(sc)
    oracle_res = [[ora_align(O.Grid(t, op), s_, G) for t, s_, G in sc] for sc in batches]
    if oracle_only:
        return
    pitch = (max(max(len(t), len(s_)) for sc in batches for t, s_, _ in sc) + 63) // 64 * 64
    dev = torch.device("cuda:0")
    bufs = []
    for sc in batches:
        T = torch.zeros(n_pairs, 3, pitch, device=dev); S = torch.zeros(n_pairs, 3, pitch, device=dev)
        for b, (t, s_, _) in enumerate(sc):
            T[b, :, :len(t)] = torch.from_numpy(t.T.copy()).to(dev); S[b, :, :len(s_)] = torch.from_numpy(s_.T.copy()).to(dev)
        bufs.append((T, S, [len(t) for t, _, _ in sc], [len(s_) for _, s_, _ in sc],
                     np.ascontiguousarray(np.stack([G.T.reshape(16) for _, _, G in sc]), np.float32)))
    torch.cuda.synchronize()
    eng = ndt.Engine(gp)
    try:
        eng.set_option(ndt.OPT_STREAM_THRESHOLD, thresh)
        eng.stream_begin(nctx, n_pairs, pitch, pitch)
        ids, got = [], []
        for T, S, tc, scn, G in bufs:
            if len(ids) - len(got) >= nctx:
                got.append(eng.stream_collect(ids[len(got)], n_pairs))
            ids.append(eng.stream_submit(T.data_ptr(), tc, pitch, S.data_ptr(), scn, pitch, G))
        while len(got) < len(ids):
            got.append(eng.stream_collect(ids[len(got)], n_pairs))
        eng.stream_end()
        for j, (res, ores, sc) in enumerate(zip(got, oracle_res, batches)):
            for b in range(n_pairs):
                compare(res[b], ores[b], f"stream[{j}][{b}] ctx={nctx} thresh={thresh}", fails, ctx, lambda j=j, b=b: ora_align(O.Grid(batches[j][b][0], op), batches[j][b][1], batches[j][b][2]))
    finally:
        eng.close()


def run_case(case, seed, fails, oracle_only=False):
    rng = np.random.default_rng([seed, case])
    kw = params(rng)
    gp, op = ndt.default_params(**kw), O.default_params(**kw)
    path = rng.choice(["single", "single_latency", "batch", "batch_latency", "sequence"], p=[0.25, 0.2, 0.25, 0.2, 0.1])
    if STREAM_P > 0 and rng.random() < STREAM_P:
        return run_stream(case, rng, kw, fails, oracle_only)
    if path == "sequence":
        return run_sequence(case, rng, kw, fails, oracle_only)
    if rng.random() < AUX_P:
        run_aux(case, rng, kw, fails, oracle_only)
    n_pairs = 1 if path.startswith("single") else int(rng.integers(2, 6))
    scenes = [scene(rng, case) for _ in range(n_pairs)]
    ctx = dict(case=case, path=str(path), **kw)
    STATS["paths"][str(path)] = STATS["paths"].get(str(path), 0) + 1
    mk = f"v{kw['variant']}_m{kw['neighbor_mode']}"
    STATS["modes"][mk] = STATS["modes"].get(mk, 0) + 1
    STATS["mt_live"] += int(kw["step_size"] <= kw["trans_epsilon"] / 2)
    if oracle_only:
        for t, s, G in scenes:
            ora_align(O.Grid(t, op), s, G)
        return
    eng = ndt.Engine(gp)
    try:
        if "latency" in path:
            eng.set_latency_mode(True)
        grids = [O.Grid(t, op) for t, _, _ in scenes]
        oracle_res = [ora_align(g, s, G) for g, (_, s, G) in zip(grids, scenes)]
        if n_pairs == 1:
            tgt, src, G = scenes[0]
            eng.set_target(tgt)
            eng.set_source(src)
            # voxel grid and one sweep, as tests/test_gpu_parity.py checks them
            from test_gpu_parity import check_voxels, check_sweep
            if not grids[0].ok:                   # no grid (guard): the engine says so per pair, and aligns like an empty target
                STATS["no_grid_cases"] = STATS.get("no_grid_cases", 0) + 1
                try:
                    eng.get_grid()
                    fails.append(dict(ctx, what="grid guard: the oracle has no grid, the engine has one"))
                except ndt.NDTError as e:
                    if e.code != -4:
                        fails.append(dict(ctx, what="grid guard", code=e.code))
                compare(eng.align(G), oracle_res[0], "align (no grid)", fails, ctx)
                return
            try:
                if not ARITH:                     # (tolerance arithmetic: tree leaf sums -- same leaves, sums to 1e-16, not bit for bit: tests/test_tolerance_mode.py)
                    STATS["voxel_checks"] += 1
                    STATS["searchable_leaves"] += int(check_voxels(eng, grids[0]) or 0)
            except AssertionError as e:
                fails.append(dict(ctx, what="voxels", detail=str(e)[:200]))
            p = O.se3_log(G.astype(np.float64)) + np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.01, 3)])
            try:
                STATS["sweep_checks"] += 1
                if ARITH:                         # the tolerance arithmetic's bar for one sweep: 1e-5 of the largest entry, the same hits
                    (s1, g1, H1, h1), (s0, g0, H0, h0) = eng.derivatives(p), O.derivatives_at(grids[0], src, p)
                    assert h1 == h0 and abs(s1 - s0) <= 1e-5 * max(abs(s0), 1e-300), ("score", s1, s0, h1, h0)
                    assert np.max(np.abs(g1 - g0)) <= 1e-5 * max(np.max(np.abs(g0)), 1e-300) and np.max(np.abs(H1 - H0)) <= 1e-5 * max(np.max(np.abs(H0)), 1e-300), "g / H"
                else:
                    check_sweep(eng.derivatives(p), O.derivatives_at(grids[0], src, p))
            except AssertionError as e:
                fails.append(dict(ctx, what="sweep", detail=str(e)[:200]))
            compare(eng.align(G), oracle_res[0], "align", fails, ctx, lambda: ora_align(grids[0], src, G))
        else:
            eng.batch_reserve(n_pairs, max(len(t) for t, _, _ in scenes), max(len(s) for _, s, _ in scenes))
            for b, (t, s, _) in enumerate(scenes):
                eng.batch_set_target(b, t)
                eng.batch_set_source(b, s)
            eng.batch_build_targets()
            res = eng.batch_align(np.stack([G for _, _, G in scenes]))
            for b in range(n_pairs):
                compare(res[b], oracle_res[b], f"batch_align[{b}]", fails, ctx, lambda b=b: ora_align(grids[b], scenes[b][1], scenes[b][2]))
    finally:
        eng.close()


def summarize(paths, out_path):
    """Merge per-seed result files into one committed summary (profiles/)."""
    runs = [json.load(open(p)) for p in paths]
    tot = {"aligns": 0, "iterations": 0, "zero_hit_aligns": 0, "hit_iteration_cap": 0, "mt_live": 0, "voxel_checks": 0, "sweep_checks": 0, "searchable_leaves": 0,
           "prefilter_checks": 0, "aux_align_checks": 0, "no_grid_cases": 0}
    paths_n, modes_n, fails = {}, {}, []
    for r in runs:
        for k in tot:
            tot[k] += r["stats"].get(k, 0)
        for k, v in r["stats"]["paths"].items():
            paths_n[k] = paths_n.get(k, 0) + v
        for k, v in r["stats"]["modes"].items():
            modes_n[k] = modes_n.get(k, 0) + v
        for f in r["failures"]:
            c = f.get("classification", {})
            fails.append({"seed": r["seed"], "case": f["case"], "path": f["path"], "what": f["what"], "variant": f.get("variant"), "neighbor_mode": f.get("neighbor_mode"),
                          "resolution": f.get("resolution"), "max_iterations": f.get("max_iterations"), "iterations_hip_oracle": f.get("it"),
                          "dtrans_m": f.get("dtrans"), "drot_rad": f.get("drot"), "hits_last": f.get("hits_last"), "detail": f.get("detail"),
                          "oracle_variant_moves_the_oracle": c.get("order_sensitive"), "hip_equals_oracle_variant": c.get("hip_equals_oracle_variant"),
                          "max_f32_ulps": f.get("max_f32_ulps"),
                          "oracle_variants": {k: ({"iterations": v["it"], "d_vs_canonical": v["d_vs_canonical"]} if "it" in v else v) for k, v in c.get("variants", {}).items()}})
    out = {"what": "tools/fuzz_parity.py: randomised differential test, HIP path (C-ABI) vs oracle; same bars as tests/test_gpu_parity.py (voxels exact, sweep rtol 1e-11, "
                   "align: iterations / converged / sweeps equal, pose inside 1e-4 m and 1e-5 rad)",
           "seeds": [r["seed"] for r in runs], "cases": sum(r["cases_run"] for r in runs), "totals": tot, "entry_paths": paths_n,
           "variant_mode_counts (v = variant, m = pclomp::NeighborSearchMethod value)": modes_n,
           "errors": sum(len(r["errors"]) for r in runs), "failures": len(fails),
           "failures_not_explained_by_the_oracles_own_order_sensitivity": sum(r.get("unexplained", 0) for r in runs),
           "failing_aligns": fails}
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"{out['cases']} cases, {tot['aligns']} aligns, {len(fails)} failing aligns, {out['failures_not_explained_by_the_oracles_own_order_sensitivity']} unexplained -> {out_path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarize", nargs="+", help="merge result files (gpurun_out/fuzz/fuzz_*.json) into --out and exit")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_fuzz.json"))
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--oracle-only", action="store_true", help="CPU dry run: scenes + oracle only")
    ap.add_argument("--aux", type=float, default=None, help="probability of the prefilter / fitness / output-cloud checks per case (default 0.15)")
    ap.add_argument("--seconds", type=float, default=1e9, help="stop after this much wall time")
    ap.add_argument("--arith", type=int, default=0, choices=[0, 1], help="1: every engine in the tolerance arithmetic (MI355NDT_OPT_ARITH); aligns are held to the same bar "
                                                                             "(the oracle's iteration count, 1e-4 m / 1e-5 rad), voxels and sweeps to the mode's own (see the code)")
    ap.add_argument("--stream", type=float, default=0.0, help="probability that a case goes through mi355ndt_stream_* (three ragged batches, 2..4 contexts)")
    a = ap.parse_args()
    if a.summarize:
        return summarize(a.summarize, a.out)
    global AUX_P, STREAM_P, ARITH
    STREAM_P = a.stream
    ARITH = a.arith
    if ARITH:
        os.environ["MI355NDT_ARITH"] = "1"      # (read by mi355ndt_create: every engine of the run)
    if a.aux is not None:
        AUX_P = a.aux
    fails, errors, done = [], [], 0
    t0 = time.time()
    for case in range(a.first, a.first + a.cases):
        if time.time() - t0 > a.seconds:
            break
        try:
            run_case(case, a.seed, fails, a.oracle_only)
        except ndt.NDTError as e:                # an error code is a legitimate outcome only if the oracle agrees there is nothing to align
            errors.append(dict(case=case, error=str(e)[:200]))
        done += 1
    out = dict(seed=a.seed, first=a.first, arith=a.arith, cases_run=done, seconds=round(time.time() - t0, 1), failures=fails, errors=errors, stats=STATS)
    od = os.path.join(ROOT, "gpurun_out", "fuzz")
    os.makedirs(od, exist_ok=True)
    json.dump(out, open(os.path.join(od, f"fuzz_{a.seed}_{a.first}.json"), "w"), indent=1)
    for f in fails:
        print("FAIL", json.dumps(f))
    for e in errors:
        print("ERROR", json.dumps(e))
    print("stats", json.dumps(STATS))
    unexplained = [f for f in fails if not (f.get("classification", {}).get("order_sensitive") or f.get("classification", {}).get("hip_equals_oracle_variant"))]
    out["unexplained"] = len(unexplained)
    json.dump(out, open(os.path.join(od, f"fuzz_{a.seed}_{a.first}.json"), "w"), indent=1)
    print(f"fuzz: {done} cases, {len(fails)} failures ({len(unexplained)} not explained by the oracle's own order sensitivity), {len(errors)} errors, {out['seconds']} s")


if __name__ == "__main__":
    main()
