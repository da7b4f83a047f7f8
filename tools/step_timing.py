"""Times one bench step (build targets + align) with the engine's event profiling off and on, and the two halves apart."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth

B, NAZ = int(os.environ.get("PAIRS", 271)), 1024
dev = torch.device("cuda:0")
N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b % 16, NAZ, device=dev)
    T[b] = t.T; S[b] = s.T
eng = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=64), device=0)
eng.batch_bind_device(T.data_ptr(), [N] * B, N, S.data_ptr(), [N] * B, N)
G = synth.default_guess()
guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (B, 16)), dtype=np.float32)
res = (ndt.Result * B)()

def timed(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def step():
    eng.batch_build_targets(); eng.batch_align_raw(guesses, res)

for prof in (False, True, False):
    eng.profile_enable(prof)
    print(f"prof={prof}: build {timed(eng.batch_build_targets):.3f} ms, align {timed(lambda: eng.batch_align_raw(guesses, res)):.3f} ms, step {timed(step):.3f} ms", flush=True)
eng.close()
