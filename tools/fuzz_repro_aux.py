#!/usr/bin/env python3
"""Replays the incremental-transform check of one `aux` case of tools/fuzz_parity.py (GPU box; test infrastructure).
  python tools/fuzz_repro_aux.py <seed> <case>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity as F                     # noqa: E402
from lv_slam_amd import ndt                 # noqa: E402
from oracle import oracle_py as O           # noqa: E402

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, case])
kw = F.params(rng)
path = rng.choice(["single", "single_latency", "batch", "batch_latency", "sequence"], p=[0.25, 0.2, 0.25, 0.2, 0.1])
assert path != "sequence"
rng.random()                                # the aux draw of run_case
tgt, src, G = F.scene(rng, case)            # run_aux's own scene
print("case", seed, case, path, kw, len(tgt), len(src), "async env", os.environ.get("MI355NDT_ASYNC"))
op = O.default_params(**kw)
eng = ndt.Engine(ndt.default_params(**kw))
eng.set_target(tgt); eng.set_source(src)
r = eng.align(G)
ro = F.ora_align(O.Grid(tgt, op), src, G)
print("iterations", r["iterations"], ro["iterations"], "final equal", np.array_equal(np.asarray(r["final"]), np.asarray(ro["final"])), "hits_last", r["hits_last"], ro["hits_last"])
a, b = eng.get_incremental()
print("transformation equal", np.array_equal(a, ro["transformation"], equal_nan=True), "previous equal", np.array_equal(b, ro["previous_transformation"], equal_nan=True))
np.set_printoptions(precision=9, linewidth=200)
print("gpu transformation\n", a, "\nora\n", np.asarray(ro["transformation"]))
print("gpu previous\n", b, "\nora\n", np.asarray(ro["previous_transformation"]))
