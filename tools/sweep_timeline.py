"""Where a sweep work item spends its time: per-phase shader-clock totals from a library built with -DNDT_TIMELINE
(hipcc <the flags of __graft_entry__.HIP_FLAGS> -DNDT_TIMELINE -DNDT_SINGLE_TU lv_slam_amd/csrc/mi355_ndt.hip -o lv_slam_amd/libexp_tl.so;
MI355NDT_LIB=<that file>.  -DNDT_SINGLE_TU: the library normally has two translation units -- the ORD = 1 kernel instantiations live in
mi355_ndt_ord1.hip -- and the timeline counters (`__device__ g_tl`) may only be defined once: build everything from the one file).
MODE=direct1|direct7 VARIANT=omp|pca PAIRS=271 AZIMUTH=1024 RESOLUTION=1.0.  The shipped library has no such hook."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth
B, NAZ = int(os.environ.get("PAIRS", 271)), int(os.environ.get("AZIMUTH", 1024))
RES = float(os.environ.get("RESOLUTION", 1.0))
MODE = {"direct7": ndt.DIRECT7, "direct1": ndt.DIRECT1}[os.environ.get("MODE", "direct7")]
VAR = 1 if os.environ.get("VARIANT", "omp") == "pca" else 0
dev = torch.device("cuda:0")
N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b, NAZ, device=dev)
    T[b] = t.T; S[b] = s.T
eng = ndt.Engine(ndt.default_params(resolution=RES, trans_epsilon=0.01, max_iterations=int(os.environ.get("MAXIT", 0)), neighbor_mode=MODE, variant=VAR), device=0)
eng.batch_bind_device(T.data_ptr(), [N] * B, N, S.data_ptr(), [N] * B, N)
G = synth.default_guess()
guesses = np.ascontiguousarray(np.broadcast_to(G.T.reshape(1, 16), (B, 16)), dtype=np.float32)
res = (ndt.Result * B)()
eng.batch_build_targets()
for _ in range(3): eng.batch_align_raw(guesses, res)
lib = ndt.load_library()
out = (ctypes.c_ulonglong * 16)()
lib.mi355ndt_debug_timeline(out)
eng.profile_enable(True); eng.profile_reset()
R = 5
for _ in range(R): eng.batch_align_raw(guesses, res)
p = eng.profile_get()
lib.mi355ndt_debug_timeline(out)
v = np.array(list(out), dtype=np.float64)
names = ["claim/loop", "setup+pt issue", "pt wait+transform", "bitmap issue", "bitmap wait+push", "drain(eval)", "reduce+write", "(items)", "row drain+arrive", "update: write-back + publish", "ticket wait", "-", "update: state + rows", "update: solve", "update: Newton step", "update: deferred re-basing"]
tot = v[:7].sum() + v[8:11].sum() + v[12:16].sum()
print(MODE, VAR, "items", int(v[7]), "sweep ms/align", p["sweep_ms"] / R, "launches", p["sweep_launches"] / R)
for n, x in [(a, c) for k, (a, c) in enumerate(zip(names, v)) if k not in (7, 11)]: print(f"  {n:22s} {x / v[7]:10.0f} cyc/item  {100 * x / tot:5.1f} %")
print("  total cyc/item", tot / v[7])
eng.close()
