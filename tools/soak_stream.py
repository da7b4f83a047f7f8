#!/usr/bin/env python3
"""Soak of the stream mode (GPU box; measurement / test infrastructure): three distinct batches are streamed through one engine again and
again (mi355ndt_stream_begin / _submit / _collect) and every word of every result record is compared with the SYNCHRONOUS
align of the same batch (batch_build_targets + batch_align, taken once up front).  Which pairs a launch hands to the next one, in which
launch a pair finishes, which wave updates it -- all of that changes from pass to pass; the bits must not.
  python tools/soak_stream.py [seconds per configuration] [contexts]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lv_slam_amd import ndt, synth          # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NB = 3
dev = torch.device("cuda:0")
CONFIGS = [("config 3: ndt_omp, 1.0 m, DIRECT7", 271, 1024, dict(variant=0, neighbor_mode=ndt.DIRECT7, resolution=1.0)),
           ("nodelet: ndt_pca, 1.0 m, DIRECT1", 271, 1024, dict(variant=1, neighbor_mode=ndt.DIRECT1, resolution=1.0)),
           ("config 5: ndt_pca, 0.5 m, DIRECT1, 131,072 points", 128, 2048, dict(variant=1, neighbor_mode=ndt.DIRECT1, resolution=0.5)),
           ("ragged: ndt_omp, 1.0 m, DIRECT7, 8,192 ... 65,536 points", 96, 1024, dict(variant=0, neighbor_mode=ndt.DIRECT7, resolution=1.0))]
out = []
for name, B, naz, kw in CONFIGS:
    n = naz * 64
    T = torch.zeros(NB, B, 3, n, device=dev)
    S = torch.zeros(NB, B, 3, n, device=dev)
    cnt = []
    for j in range(NB):
        c = []
        for k in range(B):
            t, s, _ = synth.make_pair(j * B + k, naz, device=dev)
            m = n if not name.startswith("ragged") else max(8192, n - ((j * B + k) * 977) % (n - 8192))
            T[j, k, :, :m] = t.T[:, :m]
            S[j, k, :, :m] = s.T[:, :m]
            c.append(m)
        cnt.append(c)
    torch.cuda.synchronize()
    eng = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64, **kw))
    G = synth.default_guess()
    guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (B, 16)), dtype=np.float32)
    ref = []
    for j in range(NB):                           # the synchronous results, once
        eng.batch_bind_device(T[j].data_ptr(), cnt[j], n, S[j].data_ptr(), cnt[j], n)
        eng.batch_build_targets()
        res = (ndt.Result * B)()
        eng.batch_align_raw(guesses, res)
        ref.append(bytes(memoryview(res)))
    eng.stream_begin(nctx, B, n, n)
    eng.profile_enable(True); eng.profile_reset()
    res = (ndt.Result * B)()
    ids, passes, bad = [], 0, 0
    t0 = time.time()
    k = 0
    while time.time() - t0 < seconds or ids:
        if time.time() - t0 < seconds:
            j = k % NB
            ids.append((eng.stream_submit(T[j].data_ptr(), cnt[j], n, S[j].data_ptr(), cnt[j], n, guesses), j))
            k += 1
        if len(ids) >= nctx or time.time() - t0 >= seconds:
            bid, j = ids.pop(0)
            eng.stream_collect_raw(bid, res)
            if bytes(memoryview(res)) != ref[j]:
                bad += 1
            passes += 1
    p = eng.profile_get()
    eng.stream_end()
    print(f"{name}: {B} pairs x {NB} batches through {nctx} contexts, {passes} streamed batches in {time.time() - t0:.1f} s, {bad} with a result word different from the synchronous "
          f"align's; launches {p['stream_launches']}, pairs handed over {p['stream_carried']}, batches re-run {p['stream_redone']}, launches that gave up {p['async_fallbacks']}")
    out.append((name, passes, bad))
    eng.close()
sys.exit(1 if any(b for _, _, b in out) else 0)
