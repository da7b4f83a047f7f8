"""Where the leaf sums of a target build spend their time: per-phase shader-clock totals from a library built with -DNDT_TIMELINE
-DNDT_SINGLE_TU (see tools/sweep_timeline.py for the build line), loaded through MI355NDT_LIB.  PAIRS=271 AZIMUTH=1024 RESOLUTION=1.0.
The shipped library has no such hook.  (Round 5's experimental seven-leaves-per-wave kernel -- docs/patches/r05_leafsum7.patch -- carries
the same stamps: LEAF7=1 labels its phases.)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lv_slam_amd import ndt, synth
B, NAZ = int(os.environ.get("PAIRS", 271)), int(os.environ.get("AZIMUTH", 1024))
RES = float(os.environ.get("RESOLUTION", 1.0))
dev = torch.device("cuda:0")
N = NAZ * 64
T = torch.empty(B, 3, N, device=dev); S = torch.empty(B, 3, N, device=dev)
for b in range(B):
    t, s, _ = synth.make_pair(b, NAZ, device=dev)
    T[b] = t.T; S[b] = s.T
eng = ndt.Engine(ndt.default_params(resolution=RES, trans_epsilon=0.01, max_iterations=0, neighbor_mode=ndt.DIRECT7, variant=0), device=0)
eng.batch_bind_device(T.data_ptr(), [N] * B, N, S.data_ptr(), [N] * B, N)
for _ in range(3): eng.batch_build_targets()
lib = ndt.load_library()
out = (ctypes.c_ulonglong * 16)()
lib.mi355ndt_debug_leaf_timeline(out)
eng.profile_enable(True); eng.profile_reset()
R = 5
for _ in range(R): eng.batch_build_targets()
p = eng.profile_get()
lib.mi355ndt_debug_leaf_timeline(out)
v = np.array(list(out), dtype=np.float64)
seven = os.environ.get("LEAF7", "0") == "1"
names = (["wait: rows + ids", "terms -> LDS", "stores + row requests", "deal leaves + id requests", "sums", "hand-down (key wait)"] if seven else
         ["wait: keys + ids", "wait: rows", "terms -> LDS", "sums", "stores + loop"])
unit = "trip" if seven else "chunk"
n_u, n_leaves, n_waves = v[8] / R, v[9] / R, v[10] / R          # (of the sampled waves)
print(f"{'seven leaves per wave' if seven else 'one wave per leaf'}: build {p['build_ms'] / R:.3f} ms; per build: {n_u:.0f} {unit}s, {n_leaves:.0f} leaves, {n_waves:.0f} waves, "
      f"mean wave lifetime {v[11] / v[10]:.0f} cycles, {unit}s per wave {v[8] / v[10]:.1f}")
tot = v[:len(names)].sum()
for k, n in enumerate(names): print(f"  {n:28s} {v[k] / v[8]:9.0f} cyc/{unit}  {100 * v[k] / tot:5.1f} %")
print(f"  total {tot / v[8]:.0f} cyc/{unit}")
eng.close()
