"""Latency mode on a drive: mi355ndt_sequence_run over N frames of 65,536 points (the nodelet's configuration); prints the per-frame
track time.  Under `rocprofv3 --kernel-trace` + tools/seq_kernels.py it gives the per-kernel picture of one round."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lv_slam_amd import ndt, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
scans, _ = synth.make_sequence(n, 1024, device="cuda")
scans = [s.cpu().numpy() for s in scans]
stamps = [0.1 * k for k in range(n)]
eng = ndt.Engine(ndt.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=ndt.VARIANT_PCA))
eng.sequence_run(scans[:8], stamps[:8])
for rep in range(3):
    t0 = time.perf_counter()
    out, st = eng.sequence_run(scans, stamps)
    wall = time.perf_counter() - t0
    its = np.mean([f["iterations"] for f in out[1:]])
    print(f"frames {n}: wall {1e3 * wall:.2f} ms, upload {st['upload_ms']:.2f}, build {st['build_ms']:.3f}, track {st['track_ms']:.3f} ms = {st['track_ms'] / (n - 1):.4f} ms/frame, "
          f"mean iterations {its:.2f}, update launches {st['update_launches']}, per round {1e3 * st['track_ms'] / st['update_launches']:.2f} us", flush=True)

# with a -DNDT_TIMELINE -DNDT_SINGLE_TU build loaded through MI355NDT_LIB: where an update kernel's cycles go (ndt_sequence.hpp stamps)
lib = ndt.load_library()
if hasattr(lib, "mi355ndt_debug_timeline"):
    import ctypes
    tl = (ctypes.c_ulonglong * 16)()
    if lib.mi355ndt_debug_timeline(tl) == 0 and tl[14]:
        names = {8: "run position (first round trip)", 9: "state + rows (second round trip, reduction, barrier)", 10: "Newton step incl. waiting for the solve",
                 11: "call-site policy (frames' ends only)", 12: "state write-back", 13: "the solve on wave 1 (runs beside wave 0)"}
        n_upd = tl[14]
        print(f"k_seq_update phases over {n_upd} updates (cycles per update, 100 MHz-independent shader clock):")
        for k in (8, 9, 10, 11, 12, 13):
            print(f"  {names[k]:60s} {tl[k] / n_upd:9.0f}")
