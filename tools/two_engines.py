"""How much does overlapping two half-batches on one GPU recover?  271 pairs as one engine vs two engines (136 + 135 pairs) driven
from two host threads, each step = build targets + align."""
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth

B, NAZ, STEPS = 271, 1024, 20
dev = torch.device("cuda:0")
N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b, NAZ, device=dev)
    T[b] = t.T; S[b] = s.T
torch.cuda.synchronize()
G = synth.default_guess()


def make(lo, hi):
    n = hi - lo
    e = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64), device=0)
    e.batch_bind_device(T[lo:hi].data_ptr(), [N] * n, N, S[lo:hi].data_ptr(), [N] * n, N)
    g = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (n, 16)), dtype=np.float32)
    return e, g, (ndt.Result * n)()


def run(parts):
    engs = [make(lo, hi) for lo, hi in parts]

    def loop(k, steps):
        e, g, r = engs[k]
        for _ in range(steps):
            e.batch_build_targets(); e.batch_align_raw(g, r)

    for k in range(len(engs)): loop(k, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if len(engs) == 1 and os.environ.get("MAIN_THREAD", "1") == "1":
        loop(0, STEPS)                            # the way bench.py drives one engine
    else:
        th = [threading.Thread(target=loop, args=(k, STEPS)) for k in range(len(engs))]
        for t in th: t.start()
        for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e, _, _ in engs: e.close()
    return B * STEPS / dt

print(f"one engine, 271 pairs (main thread): {run([(0, B)]):9.0f} registrations/s")
os.environ["MAIN_THREAD"] = "0"
print(f"one engine, 271 pairs (own thread) : {run([(0, B)]):9.0f} registrations/s")
print(f"two engines, 136 + 135 pairs: {run([(0, 136), (136, B)]):9.0f} registrations/s")
print(f"three engines               : {run([(0, 91), (91, 181), (181, B)]):9.0f} registrations/s")
