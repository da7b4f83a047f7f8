#!/usr/bin/env python3
"""GPU box: what the tolerance arithmetic (MI355NDT_OPT_ARITH = 1) costs in accuracy, pair by pair, against the CPU oracle -- over EVERY pair of a job
(BASELINE config 4: pairs 0..4540; config 5's share: 0..127), not a sample.  For every pair: oracle align, GPU align in the exact arithmetic, GPU align
in the tolerance arithmetic; reported: iteration flips, pairs beyond north_star's tolerance (trans < 1e-4 m, rot < 1e-5 rad), largest / median deltas, for
tolerance-vs-oracle, exact-vs-oracle and tolerance-vs-exact.   usage: tools/tolerance_study.py --pairs 4541 [--variant pca --mode direct1 --resolution 0.5 --azimuth 2048] > profiles/r06_tolerance_<cfg>.json"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lv_slam_amd import ndt, synth
from lv_slam_amd import dist as shard
from oracle import oracle_py as O

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="omp"); ap.add_argument("--mode", default="direct7"); ap.add_argument("--resolution", type=float, default=1.0)
ap.add_argument("--azimuth", type=int, default=1024); ap.add_argument("--pairs", type=int, default=4541); ap.add_argument("--chunk", type=int, default=271)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
ctx = bench.Ctx()
ctx.rank, ctx.world, ctx.local, ctx.dev, ctx.dist, ctx.use_dist = 0, 1, 0, torch.device("cuda", 0), None, False
ctx.on_dev, ctx.shard, ctx.ndt, ctx.G = True, shard, ndt, synth.default_guess()
torch.cuda.set_device(0)
prm = ndt.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=bench.MODES[a.mode], variant=1 if a.variant == "pca" else 0)
op = O.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=bench.MODES[a.mode], variant=1 if a.variant == "pca" else 0)
threads = bench.cpu_quota() or os.cpu_count() or 8
O.lib().ora_set_threads(int(threads))
engs = {}
for name, arith in (("exact", 0), ("tolerance", 1)):
    engs[name] = ndt.Engine(prm, device=0)
    engs[name].set_option(ndt.OPT_ARITH, arith)
rows = {k: [] for k in ("tolerance_vs_oracle", "exact_vs_oracle", "tolerance_vs_exact")}
its = {k: [] for k in ("oracle", "exact", "tolerance")}
t_cpu = 0.0
t0 = time.perf_counter()
for first in range(0, a.pairs, a.chunk):
    nb = min(a.chunk, a.pairs - first)
    W, _ = bench.generate_synthetic(ctx, synth, list(range(first, first + nb)), a.azimuth)
    guesses = np.ascontiguousarray(np.broadcast_to(ctx.G.T.reshape(1, 16), (nb, 16)), dtype=np.float32)
    R = {}
    for name, eng in engs.items():
        eng.batch_bind_device(W["T"].data_ptr(), W["tcnt"], W["pitch"], W["S"].data_ptr(), W["scnt"], W["pitch"])
        eng.batch_build_targets()
        res = (ndt.Result * nb)()
        eng.batch_align_raw(guesses, res)
        R[name] = np.frombuffer(np.frombuffer(res, dtype=np.uint8).copy(), dtype=bench.RES_DT)
    for k in range(nb):
        fe, ft = R["exact"]["final"][k].reshape(4, 4).T, R["tolerance"]["final"][k].reshape(4, 4).T
        rows["tolerance_vs_exact"].append(bench.se3_err(fe, ft) + (int(R["exact"]["it"][k] != R["tolerance"]["it"][k]), int(R["exact"]["conv"][k] != R["tolerance"]["conv"][k]), int(R["tolerance"]["status"][k] == 1)))
        its["exact"].append(int(R["exact"]["it"][k])); its["tolerance"].append(int(R["tolerance"]["it"][k]))
        if not a.no_oracle:
            c0 = time.perf_counter()
            ro = O.align(O.Grid(bench.cloud_np(W, "T", k), op), bench.cloud_np(W, "S", k), ctx.G)
            t_cpu += time.perf_counter() - c0
            its["oracle"].append(int(ro["iterations"]))
            for nm, f, r in (("tolerance_vs_oracle", ft, R["tolerance"]), ("exact_vs_oracle", fe, R["exact"])):
                rows[nm].append(bench.se3_err(ro["final"], f) + (int(ro["iterations"] != int(r["it"][k])), int(bool(ro["converged"]) != bool(r["conv"][k])), int(r["status"][k] == 1)))
    del W
    print(f"pairs {first}..{first + nb - 1} done, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)

def summary(v):
    if not v:
        return None
    v = np.array(v, dtype=np.float64)
    beyond = (v[:, 0] >= 1e-4) | (v[:, 1] >= 1e-5)
    worst = int(np.argmax(v[:, 0]))
    return {"pairs": len(v), "iteration_flips": int(v[:, 2].sum()), "converged_flag_flips": int(v[:, 3].sum()), "pairs_beyond_tolerance": int(beyond.sum()),
            "pairs_beyond_tolerance_without_an_iteration_flip": int((beyond & (v[:, 2] == 0)).sum()),
            "pairs_flagged_by_the_engine": int(v[:, 4].sum()), "pairs_beyond_tolerance_and_not_flagged": int((beyond & (v[:, 4] == 0)).sum()),
            "pairs_beyond_tolerance_list": [int(k) for k in np.nonzero(beyond)[0][:16]],
            "max_dtrans_m": float(v[:, 0].max()), "max_drot_rad": float(v[:, 1].max()), "pair_of_max_dtrans": worst,
            "median_dtrans_m": float(np.median(v[:, 0])), "p99_dtrans_m": float(np.percentile(v[:, 0], 99)), "p999_dtrans_m": float(np.percentile(v[:, 0], 99.9)),
            "median_drot_rad": float(np.median(v[:, 1])), "p99_drot_rad": float(np.percentile(v[:, 1], 99)),
            "pairs_bit_identical_pose": int(((v[:, 0] == 0) & (v[:, 1] == 0)).sum())}
out = {"workload": f"pairs 0..{a.pairs - 1} of {a.azimuth * 64}-point synthetic HDL-64E scan pairs, ndt_{a.variant}, {a.resolution} m, {a.mode.upper()}, eps 0.01, max_iter 64, guess = identity + 1 m in x",
       "tolerance": "trans < 1e-4 m, rot < 1e-5 rad (north_star)", "oracle": "oracle/ndt_oracle.c, every pair" if not a.no_oracle else None,
       "oracle_threads": int(threads), "oracle_registrations_per_s": round(a.pairs / t_cpu, 2) if t_cpu > 0 else None,
       "mean_iterations": {k: round(float(np.mean(v)), 4) for k, v in its.items() if v},
       **{k: summary(v) for k, v in rows.items()}, "seconds": round(time.perf_counter() - t0, 1)}
print(json.dumps(out, indent=1), flush=True)
