"""Single-registration latency (BASELINE config 2 / the live nodelet's mode): one 65,536-pt pair through the
drop-in entry points (host AoS clouds in, PCIe included) and through the device-resident batch path with B = 1."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from lv_slam_amd import ndt, synth
tgt, src, _ = synth.make_pair(0, 1024, device="cuda")
T = tgt.T.contiguous()[None].contiguous(); S = src.T.contiguous()[None].contiguous()
tgt_h, src_h = tgt.cpu().numpy(), src.cpu().numpy()
G = synth.default_guess()
import os
LAT = bool(os.environ.get("LATENCY_MODE"))            # LATENCY_MODE=1: the opt-in fine-grained sweep (mi355ndt_set_latency_mode)
for variant, mode, name in ((0, ndt.DIRECT7, "ndt_omp/DIRECT7"), (1, ndt.DIRECT1, "ndt_pca/DIRECT1")):
    name += " [latency mode]" if LAT else ""
    prm = ndt.default_params(trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    e = ndt.Engine(prm)
    e.set_latency_mode(LAT)
    def host():
        e.set_target(tgt_h); e.set_source(src_h); return e.align(G)
    for _ in range(3): r = host()
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); r = host(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    # the nodelet's per-frame pattern against a resident target (scan_matching_odom_nodelet.cpp:220-221): setInputSource + align (+ output cloud)
    src32 = np.zeros((len(src_h), 8), np.float32); src32[:, :3] = src_h          # pcl::PointXYZI records
    def frame(cloud, out):
        e.set_source(cloud); r = e.align(G)
        if out: e.get_aligned()
        return r
    v = {}
    for nm, cloud, out in (("xyz12", src_h, False), ("xyzi32", src32, False), ("xyzi32+output", src32, True)):
        for _ in range(3): frame(cloud, out)
        w = []
        for _ in range(20):
            t0 = time.perf_counter(); frame(cloud, out); w.append(time.perf_counter() - t0)
        v[nm] = np.median(np.array(w) * 1e3)
    print(f"{name}: set_source+align against a resident target: 12-B records {v['xyz12']:.3f} ms, 32-B PointXYZI records {v['xyzi32']:.3f} ms, "
          f"with the output cloud fetched {v['xyzi32+output']:.3f} ms", flush=True)
    e2 = ndt.Engine(prm)
    e2.set_latency_mode(LAT)
    n = tgt.shape[0]
    e2.batch_bind_device(T.data_ptr(), [n], n, S.data_ptr(), [n], n)
    def dev():
        e2.batch_build_targets(); return e2.batch_align(G)[0]
    for _ in range(3): r2 = dev()
    u = []
    for _ in range(20):
        t0 = time.perf_counter(); r2 = dev(); u.append(time.perf_counter() - t0)
    u = np.array(u) * 1e3
    a = []
    for _ in range(20):
        t0 = time.perf_counter(); r3 = e2.batch_align(G)[0]; a.append(time.perf_counter() - t0)
    a = np.array(a) * 1e3
    print(f"{name}: iterations {r['iterations']}; host clouds in (set_target+set_source+align): median {np.median(t):.3f} ms "
          f"(p10 {np.percentile(t,10):.3f}, p90 {np.percentile(t,90):.3f}); device-resident build+align: {np.median(u):.3f} ms; "
          f"align only: {np.median(a):.3f} ms", flush=True)
