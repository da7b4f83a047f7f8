#!/usr/bin/env python3
"""Replays one case of tools/fuzz_parity.py and finds the Newton iteration at which the HIP path and the oracle part ways:
aligns with max_iterations = 1, 2, 3, ... on both sides, then compares the sweep (score, g, H, hits) at the last common pose.
  python tools/fuzz_repro.py <seed> <case> <pair>          (GPU box; test infrastructure)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as F                     # noqa: E402
from lv_slam_amd import ndt                 # noqa: E402
from oracle import oracle_py as O           # noqa: E402
from conftest import se3_err                # noqa: E402

seed, case, pair = (int(v) for v in sys.argv[1:4])
rng = np.random.default_rng([seed, case])
kw = F.params(rng)
path = rng.choice(["single", "single_latency", "batch", "batch_latency", "sequence"], p=[0.25, 0.2, 0.25, 0.2, 0.1])
n_pairs = 1 if path.startswith("single") else int(rng.integers(2, 6))
scenes = [F.scene(rng, case) for _ in range(n_pairs)]
t, s, G = scenes[pair]
print("case", seed, case, pair, path, kw, "n_tgt", len(t), "n_src", len(s))
last_common = G
for k in range(1, kw["max_iterations"] + 1):
    kk = dict(kw, max_iterations=k)
    gp, op = ndt.default_params(**kk), O.default_params(**kk)
    eng = ndt.Engine(gp)
    eng.set_target(t); eng.set_source(s)
    grid = O.Grid(t, op)
    r, ro = eng.align(G), O.align(grid, s, G)
    dt, dr = se3_err(ro["final"], r["final"])
    print(f"cap {k:3d}: it {r['iterations']:3d}/{ro['iterations']:3d} hits {r['hits_last']:6d}/{ro['hits_last']:6d} score {r['score']:.17g} / {ro['score']:.17g}  d=({dt:.3e}, {dr:.3e})")
    if dt > 1e-9 or dr > 1e-9 or r["iterations"] != ro["iterations"]:
        p = O.se3_log(np.asarray(last_common, np.float64))
        a, b = eng.derivatives(p), O.derivatives_at(grid, s, p)
        print("sweep at the last common pose: hits", a[3], b[3], "score", a[0], b[0])
        print(" g gpu", np.asarray(a[1])); print(" g ora", np.asarray(b[1]))
        print(" |dH| max", np.abs(np.asarray(a[2]) - np.asarray(b[2])).max(), "|H| max", np.abs(np.asarray(b[2])).max())
        ev = np.linalg.eigvalsh((np.asarray(b[2]) + np.asarray(b[2]).T) / 2)
        print(" eig(H) oracle", ev)
        eng.close()
        break
    last_common = ro["final"]
    eng.close()
