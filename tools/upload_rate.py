"""Host-cloud upload rate of one engine (batch_set_source only): staging threads vs clouds per second."""
import sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
from lv_slam_amd import ndt
B, N = 64, 65536
for rec in (8, 3):
    a = np.random.default_rng(0).normal(size=(B, N, rec)).astype(np.float32)
    stride = rec * 4
    e = ndt.Engine(ndt.default_params())
    e.batch_reserve(B, N, N)
    for nt in (1, 2, 4, 8, 16):
        pool = ThreadPoolExecutor(nt)
        def up(w):
            for k in range(w, B, nt):
                e.batch_set_source_raw(k, a.ctypes.data + k * N * stride, N, stride)
        list(pool.map(up, range(nt))); e.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            list(pool.map(up, range(nt)))
        t1 = time.perf_counter()
        e.batch_build_targets() if False else None
        e.synchronize()
        dt = time.perf_counter() - t0
        print(f"record {stride} B, {nt:2d} threads: {4 * B / dt:8.0f} clouds/s ({4 * B * N * 12 / dt / 1e9:5.1f} GB/s over PCIe, staging calls returned after {1e3 * (t1 - t0):.1f} ms of {1e3 * dt:.1f} ms)", flush=True)
        pool.shutdown()
    e.close()
