"""Would overlapping the target build with sweeps (or two half builds with each other) pay?  Two engines on one GPU from two
host threads: (1) one engine builds 271 targets alone, (2) two engines build 136 + 135 targets at the same time, (3) engine A
aligns 136 pairs against resident grids while engine B builds 135 targets, each also timed alone."""
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth
B, NAZ = 271, 1024
dev = torch.device("cuda:0"); N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b, NAZ, device=dev); T[b] = t.T; S[b] = s.T
prm = ndt.default_params(trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
G = synth.default_guess()
def engine(lo, hi):
    e = ndt.Engine(prm, device=0); n = hi - lo
    e.batch_bind_device(T[lo:hi].data_ptr(), [N] * n, N, S[lo:hi].data_ptr(), [N] * n, N)
    g = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (n, 16)), dtype=np.float32)
    r = (ndt.Result * n)()
    e.batch_build_targets(); e.batch_align_raw(g, r)
    return e, g, r
def timed(fns, reps=20):
    out = [0.0] * len(fns)
    bar = threading.Barrier(len(fns) + 1)
    def run(i):
        bar.wait(); t0 = time.perf_counter()
        for _ in range(reps): fns[i]()
        torch.cuda.synchronize(); out[i] = (time.perf_counter() - t0) / reps * 1e3
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in th: t.start()
    bar.wait()
    for t in th: t.join()
    return [round(x, 3) for x in out]
full, gf, rf = engine(0, B)
ea, ga, ra = engine(0, 136)
eb, gb, rb = engine(136, B)
def build_sync(e): e.batch_build_targets(); torch.cuda.synchronize()
print("build 271 alone ms:", timed([lambda: build_sync(full)]))
print("build 136 alone ms:", timed([lambda: build_sync(ea)]), " build 135 alone ms:", timed([lambda: build_sync(eb)]))
print("build 136 || build 135 ms:", timed([lambda: build_sync(ea), lambda: build_sync(eb)]))
print("align 136 alone ms:", timed([lambda: ea.batch_align_raw(ga, ra)]))
print("align 136 || build 135 ms:", timed([lambda: ea.batch_align_raw(ga, ra), lambda: build_sync(eb)]))
print("step (build+align) 271 alone ms:", timed([lambda: (full.batch_build_targets(), full.batch_align_raw(gf, rf))]))
# (4) two engines over full batches, software-pipelined across steps: only one of them sweeps at a time (a lock around align),
#     the other one's target build runs underneath
full2, gf2, rf2 = engine(0, B)
lock = threading.Lock()
def step_locked(e, g, r):
    e.batch_build_targets()
    with lock:
        e.batch_align_raw(g, r)
t = timed([lambda: step_locked(full, gf, rf), lambda: step_locked(full2, gf2, rf2)], reps=20)
print("two engines, align under a lock, ms per step of EACH engine:", t, "-> %.0f reg/s" % (2 * B / (max(t) * 1e-3)))
t1 = timed([lambda: step_locked(full, gf, rf)], reps=20)
print("one engine ms per step:", t1, "-> %.0f reg/s" % (B / (t1[0] * 1e-3)))
