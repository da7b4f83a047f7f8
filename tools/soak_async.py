#!/usr/bin/env python3
"""Soak of the one-launch batch align (GPU box; measurement / test infrastructure): the same batch is built and aligned again and again,
and every word of every result record is compared with the first run's.  The hand-off between workgroups inside the launch (tickets,
write-through rows and states, last-arriver updates: DESIGN.md 4.2a) has no fixed schedule -- which wave runs which item and which one
updates a pair changes from run to run -- so equal bits over many runs is the evidence that the schedule is no part of the result.
  python tools/soak_async.py [seconds per configuration]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lv_slam_amd import ndt, synth          # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dev = torch.device("cuda:0")
CONFIGS = [("config 3: ndt_omp, 1.0 m, DIRECT7", 271, 1024, dict(variant=0, neighbor_mode=ndt.DIRECT7, resolution=1.0)),
           ("nodelet: ndt_pca, 1.0 m, DIRECT1", 271, 1024, dict(variant=1, neighbor_mode=ndt.DIRECT1, resolution=1.0)),
           ("config 5: ndt_pca, 0.5 m, DIRECT1, 131,072 points", 128, 2048, dict(variant=1, neighbor_mode=ndt.DIRECT1, resolution=0.5)),
           ("ragged: ndt_omp, 1.0 m, DIRECT7, 8,192 ... 65,536 points", 96, 1024, dict(variant=0, neighbor_mode=ndt.DIRECT7, resolution=1.0))]
out = []
for name, B, naz, kw in CONFIGS:
    n = naz * 64
    T = torch.zeros(B, 3, n, device=dev)
    S = torch.zeros(B, 3, n, device=dev)
    tc, sc = [], []
    for k in range(B):
        t, s, _ = synth.make_pair(k, naz, device=dev)
        m = n if not name.startswith("ragged") else max(8192, n - (k * 977) % (n - 8192))
        T[k, :, :m] = t.T[:, :m]
        S[k, :, :m] = s.T[:, :m]
        tc.append(m); sc.append(m)
    torch.cuda.synchronize()
    eng = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64, **kw))
    eng.batch_bind_device(T.data_ptr(), tc, n, S.data_ptr(), sc, n)
    G = synth.default_guess()
    guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (B, 16)), dtype=np.float32)
    res = (ndt.Result * B)()
    first, runs, bad = None, 0, 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        eng.batch_build_targets()
        eng.batch_align_raw(guesses, res)
        raw = bytes(memoryview(res))
        if first is None:
            first = raw
        elif raw != first:
            bad += 1
        runs += 1
    prof_ok = eng.get_option(ndt.OPT_ASYNC_ALIGN)
    its = np.frombuffer(first, dtype=np.uint8)
    print(f"{name}: {B} pairs, {runs} build+align runs in {time.time() - t0:.1f} s, {bad} runs with a result word different from the first run's (async option {prof_ok})")
    out.append((name, runs, bad))
    eng.close()
sys.exit(1 if any(b for _, _, b in out) else 0)
