"""Times full 271-pair derivative-sweep launches in isolation: batch align with max_iterations = 0 (initial sweep + one step
sweep for every pair, then the loop ends), engine event profile on.  Used to compare sweep-kernel variants."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth

B, NAZ = int(os.environ.get("PAIRS", 271)), 1024
MODE = {"direct7": ndt.DIRECT7, "direct1": ndt.DIRECT1, "direct26": ndt.DIRECT26, "kdtree": ndt.KDTREE}[os.environ.get("MODE", "direct7")]
VAR = 1 if os.environ.get("VARIANT", "omp") == "pca" else 0
dev = torch.device("cuda:0")
N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b, NAZ, device=dev)
    T[b] = t.T; S[b] = s.T
eng = ndt.Engine(ndt.default_params(trans_epsilon=0.01, max_iterations=0, neighbor_mode=MODE, variant=VAR), device=0)
eng.batch_bind_device(T.data_ptr(), [N] * B, N, S.data_ptr(), [N] * B, N)
G = synth.default_guess()
guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (B, 16)), dtype=np.float32)
res = (ndt.Result * B)()
eng.batch_build_targets()
for _ in range(3):
    eng.batch_align_raw(guesses, res)
eng.profile_enable(True); eng.profile_reset()
R = 10
for _ in range(R):
    eng.batch_align_raw(guesses, res)
p = eng.profile_get()
full = p["sweep_alg_bytes"] / 1e9
print(f"sweep total {p['sweep_ms'] / R:.3f} ms per align over {p['sweep_launches'] / R:.1f} launches; hits {p['sweep_hits'] / R / 1e6:.1f} M; "
      f"algorithmic {full / R:.2f} GB per align -> {full / (p['sweep_ms'] * 1e-3) / 1e3:.2f} TB/s")
eng.close()
