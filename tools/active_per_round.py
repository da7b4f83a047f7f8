"""How many pairs of the default bench batch are still active in each lockstep round (iteration histogram of one batch align):
the numbers behind the tail analysis in DESIGN.md section 10."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from lv_slam_amd import ndt, synth
B, NAZ = int(os.environ.get('PAIRS', 271)), int(os.environ.get('AZIMUTH', 1024))
RES = float(os.environ.get('RESOLUTION', 1.0))
MODE = {'direct7': ndt.DIRECT7, 'direct1': ndt.DIRECT1}[os.environ.get('MODE', 'direct7')]
VAR = 1 if os.environ.get('VARIANT', 'omp') == 'pca' else 0
dev = torch.device("cuda:0"); N = NAZ*64
T = torch.empty(B,3,N,device=dev); S = torch.empty(B,3,N,device=dev)
for b in range(B):
    t,s,_ = synth.make_pair(b, NAZ, device=dev); T[b]=t.T; S[b]=s.T
eng = ndt.Engine(ndt.default_params(resolution=RES, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODE, variant=VAR), device=0)
eng.batch_bind_device(T.data_ptr(), [N]*B, N, S.data_ptr(), [N]*B, N)
G = synth.default_guess()
guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1,16),(B,16)),dtype=np.float32)
res = (ndt.Result*B)()
eng.batch_build_targets(); eng.batch_align_raw(guesses,res)
it = np.array([r.iterations for r in res])
print("iterations hist", np.bincount(it))
# sweeps per pair = iterations + 1 ; active in round r (0-based) = #pairs with iterations+1 > r
print("active per round", [int((it+1 > r).sum()) for r in range(it.max()+2)])
