#!/usr/bin/env python3
"""bench.py -- NDT registrations/sec on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the hot path over one batch of synthetic scan pairs resident in HBM:
voxelise every target (setInputTarget) + align every pair (align), then -- for N > 1 -- one RCCL
all-gather of the 96-byte pose records, packed on the device by the engine (no host hop).
Pairs are sharded round-robin over ranks (pair index i -> rank i mod N), no data-path collective.

  default            : weak scaling, every rank owns `--pairs` pairs (BASELINE config 3 per GPU: 271)
  --total-pairs 4541 : BASELINE config 4 as worded -- a fixed job of T pairs over the N ranks (strong scaling,
                       shards differ by at most one pair; the shorter ones pad their record block with pair_id = -1)
  (N = 1)            : also reports the drop-in path's rate (host AoS clouds in, PCIe inclusive) as `value_host_clouds`;
                       --no-host-clouds skips that leg, --host-clouds forces it for N > 1 (rank 0)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel: the derivative
sweep, algorithmic bytes / HIP-event time, peak 8 TB/s HBM), `roofline_valu` (what actually bounds the sweep) and
`cpu_baseline` (the oracle = CPU restatement of ndt_omp, timed on this box's host cores on a bounded sample, in the
reference-shaped arrangement of SURVEY.md 8(d) and as the optimised port; the reference itself cannot be built here).
"""
import argparse
import gc
import json
import os
import sys
import time

if os.environ.get("LV_SLAM_CPUS"):                   # e.g. "0-63": keep the process (and the memory it first touches) on one socket
    lo, hi = os.environ["LV_SLAM_CPUS"].split("-")
    os.sched_setaffinity(0, range(int(lo), int(hi) + 1))
# RCCL's intra-node transport shares device buffers between the ranks' processes through HIP IPC handles.  The host driver of these
# boxes only implements the dmabuf flavour of IPC: with the HSA runtime's legacy IPC mode left on, hipIpcGetMemHandle fails with
# "invalid argument" and ncclCommInitRank / the first collective of an N > 1 run dies.  The HSA runtime reads the variable when it
# starts (first `import torch` that touches the GPU), so it is set here, before any import -- the driver's environment need not carry it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_PROC_BIND", "close")      # pinned OpenMP threads for the CPU leg (read when libgomp starts)
# ... and a libgomp that starts with OMP_PROC_BIND set binds the thread that loaded it to ONE cpu, which every thread created
# later inherits: remember what the process may use, and give it back to the threads that are not OpenMP's (apply_affinity)
ALLOWED_CPUS = set(os.sched_getaffinity(0))
PIN_CPUS = None

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)
MODES = {"direct1": 3, "direct7": 2, "direct26": 1, "kdtree": 0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: as many as make the timed region >= 0.5 s, at least 20)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=271, help="pairs per GPU per step (BASELINE config 3: 271); weak scaling")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="fixed job of this many pairs over all ranks (BASELINE config 4: 4541); strong scaling; overrides --pairs")
    ap.add_argument("--config4-pairs", type=int, default=4541,
                    help="size of the BASELINE config 4 job timed beside the default weak-scaling line (strong scaling over the same ranks; 0 = skip)")
    ap.add_argument("--azimuth", type=int, default=1024, help="firings per revolution; x64 beams = points per cloud")
    ap.add_argument("--mode", default="direct7", choices=sorted(MODES))
    ap.add_argument("--variant", default="omp", choices=["omp", "pca"])
    ap.add_argument("--resolution", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time of the parity/port sample of the cpu_baseline leg (0 = skip the leg)")
    ap.add_argument("--host-clouds", action="store_true", help="time the drop-in path (host AoS clouds in, PCIe inclusive) also when N > 1")
    ap.add_argument("--no-host-clouds", action="store_true", help="skip the drop-in path's leg (on by default in the single-GPU run)")
    ap.add_argument("--uploaders", type=int, default=0, help="staging threads per engine of the --host-clouds leg (0 = from the CPU quota)")
    ap.add_argument("--prefiltered", action="store_true",
                    help="the reference-conditioned row instead of the headline: raw scans -> device prefilter (0.5-100 m gate + 0.1 m VoxelGrid, "
                         "launch/dlo_kitti.launch:30-36) -> target / source -> align, one registration at a time (use with --azimuth 2048 --pairs 64)")
    ap.add_argument("--seq-frames", type=int, default=271, help="frames of the latency-mode leg (value_sequential; 0 = skip; single-GPU run only)")
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes per sweep launch from separate rocprofv3 --pmc passes")
    ap.add_argument("--f32-sum-order", type=int, default=0, choices=[0, 1],
                    help="MI355NDT_OPT_F32_SUM_ORDER of every engine of the run: 0 = (t0 + t1) + t2, the canonical order of the fixtures; 1 = (t0 + t2) + t1, "
                         "the lane pairing of Eigen 3.3's SSE predux -- the parity legs then check against the oracle's matching variant (ORA_VAR_SUM3_02_1)")
    ap.add_argument("--arith", type=int, default=int(os.environ.get("MI355NDT_ARITH", "0") == "1"), choices=[0, 1],
                    help="MI355NDT_OPT_ARITH of every engine of the run: 0 = the exact arithmetic (results equal the oracle's bit for bit; what `value` is measured in), "
                         "1 = tolerance arithmetic for the whole line (the default line already carries it as the `tolerance_mode` block beside the exact `value`)")
    ap.add_argument("--no-tolerance-mode", action="store_true", help="skip the `tolerance_mode` blocks (the same jobs again under MI355NDT_OPT_ARITH = 1, compared pair by pair with the exact results)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` block of the single-GPU line (the nodelet's ndt_pca / DIRECT1 and BASELINE config 5 with DIRECT7 and "
                         "DIRECT1, each timed for --other-seconds with its own roofline and oracle parity sample)")
    ap.add_argument("--other-seconds", type=float, default=0.35, help="timed region of every `other_configs` entry")
    ap.add_argument("--no-stream", action="store_true", help="skip the streamed job: `value` is then the synchronous job's rate (as in rounds 1-4)")
    ap.add_argument("--stream-contexts", type=int, default=4, help="batches resident in the engine's stream mode (2..4); four let a launch's stragglers ride through two further launches (config 5 needs that: 20.3 k against 18.9 k with three)")
    ap.add_argument("--stream-batches", type=int, default=4, help="distinct batches the streamed job rotates through")
    ap.add_argument("--stream-reserve", type=int, default=None, help="workgroup slots the stream's launches leave free for the next batch's build (MI355NDT_OPT_STREAM_RESERVE; default: the engine's)")
    ap.add_argument("--kitti-dir", default=None,
                    help="a KITTI odometry sequence's velodyne directory (<seq>/velodyne/*.bin, N x 4 f32): consecutive frames (k, k+1) become the "
                         "pairs of the run instead of the synthetic scans (scripts/lidar_odom_kitti.sh:6); clouds are ragged, `data` says \"kitti\"")
    ap.add_argument("--kitti-prefilter", action="store_true",
                    help="with --kitti-dir: every scan first goes through the device prefilter (0.5-100 m gate + 0.1 m VoxelGrid, launch/dlo_kitti.launch:30-36)")
    return ap.parse_args()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n_gpus, argv, port):
    """The command `bench.py --gpus N` turns itself into when nobody launched it as N ranks: exactly the driver's form
    (one rank per GPU on one node, rendezvous on 127.0.0.1 -- the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv, visible_devices=None, execve=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: become N ranks (torch.distributed.run) instead of
    running ONE rank that reports n_gpus = 1.  Refuses (non-zero exit) when fewer than N devices are visible -- never a silent
    smaller run.  (LV_SLAM_BENCH_BACKEND=gloo, the functional check of tests/test_bench_gpu.py, lets ranks share a GPU.)"""
    backend = os.environ.get("LV_SLAM_BENCH_BACKEND", "nccl")
    visible = torch.cuda.device_count() if visible_devices is None else visible_devices
    if visible < (a.gpus if backend == "nccl" else 1):
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {visible} GPU(s) visible to this process; refusing to run a smaller job under that name")
    cmd = launch_command(a.gpus, argv, free_port())
    env = dict(os.environ)
    env["LV_SLAM_BENCH_LAUNCHER"] = "bench.py re-executed itself under torch.distributed.run"
    print("bench: " + " ".join(cmd), file=sys.stderr, flush=True)
    (execve or os.execve)(cmd[0], cmd, env)


def se3_err(A, B):
    E = np.linalg.inv(np.asarray(A, np.float64)) @ np.asarray(B, np.float64)
    w = np.array([E[2, 1] - E[1, 2], E[0, 2] - E[2, 0], E[1, 0] - E[0, 1]]) / 2.0
    return float(np.linalg.norm(E[:3, 3])), float(np.arctan2(np.linalg.norm(w), min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1) / 2))))


def cpu_quota():
    """CPUs this process may actually use: the cgroup quota (cpu.max) when there is one, else None."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                return max(1, int(int(q) / int(per)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except Exception:
        pass
    return None


def pin_to_gpu_numa_node(dev_index):
    """Keep this process -- its staging threads and the host memory it first touches -- on the NUMA node the GPU hangs off.
    On the 2-socket hosts of this pool a staging thread that reads the clouds across the socket link halves the host-cloud rate."""
    if os.environ.get("LV_SLAM_CPUS"):
        return os.environ["LV_SLAM_CPUS"]
    try:
        from lv_slam_amd import ndt
        node = ndt.load_library().mi355ndt_host_numa_node(dev_index)      # hipDeviceAttributeHostNumaId, else the PCI device's sysfs entry
        if node < 0:                                   # containers often hide the PCI tree: ask the SMI
            import re
            import subprocess
            txt = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showtoponuma", "-d", str(dev_index)], capture_output=True, text=True, timeout=90).stdout
            m = re.search(r"Numa Node:\s*(-?\d+)", txt)
            node = int(m.group(1)) if m else -1
        if node < 0:
            print("bench: the GPU's NUMA node is unknown, process not pinned", file=sys.stderr)
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        global PIN_CPUS
        PIN_CPUS = (cpus & ALLOWED_CPUS) or set(ALLOWED_CPUS)
        os.sched_setaffinity(0, PIN_CPUS)
        return f"numa node {node} ({len(PIN_CPUS)} CPUs)"
    except Exception as e:
        print("bench: not pinned to the GPU's NUMA node:", repr(e), file=sys.stderr)
        return None


def apply_affinity():
    """(Re-)apply the process-wide CPU set to the calling thread: the GPU's NUMA node if known, else everything allowed."""
    try:
        os.sched_setaffinity(0, PIN_CPUS or ALLOWED_CPUS)
    except OSError:
        pass
    return len(os.sched_getaffinity(0))


def host_info():
    model, phys = "unknown", None
    try:
        lines = open("/proc/cpuinfo").read().split("\n")
        model = [l.split(":", 1)[1].strip() for l in lines if l.startswith("model name")][0]
        cores = {}
        pid = None
        for l in lines:
            if l.startswith("physical id"):
                pid = l.split(":")[1].strip()
            elif l.startswith("core id") and pid is not None:
                cores[(pid, l.split(":")[1].strip())] = 1
        phys = len(cores) or None
    except Exception:
        pass
    return model, phys, os.cpu_count() or 1


def baseline_config_name(a, n_points):
    """Which BASELINE.json config a weak-scaling run is (only the literal one is called by its name)."""
    if a.variant == "omp" and a.mode == "direct7" and a.resolution == 1.0 and n_points == 65536:
        return "BASELINE config 3" if a.pairs == 271 else "BASELINE config 3's workload at another batch size"
    if a.variant == "pca" and a.resolution == 0.5 and n_points == 131072:
        return "BASELINE config 5's per-GPU share" + ("" if a.mode == "direct7" else f" with {a.mode.upper()} (the nodelet's neighbour mode)")
    return "a variation of BASELINE config 3"


def static_profile(name, workload_key):
    """A measurement that cannot be taken inside the timed run (PMC counters need their own rocprofv3 passes): read the committed
    file under profiles/ -- only when it was taken on this very workload -- and say so in the JSON line."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
        if d.get("workload_key") == workload_key:
            return d, "static: profiles/" + name
    except Exception:
        pass
    return None, None


VALU_FILES = ("r06_valu.json", "r06_valu_pca_direct1.json", "r06_valu_cfg5_d1.json", "r06_valu_cfg5_d7.json", "r05_valu.json", "r05_valu_pca_direct1.json", "r05_valu_cfg5_d1.json", "r05_valu_cfg5_d7.json", "r04_valu.json", "r04_valu_pca_direct1.json",
              "r04_valu_cfg5_d1.json", "r03_valu.json", "r03_valu_pca_direct1.json", "r03_valu_cfg5_d1.json", "r02_valu.json", "r02_valu_pca_direct1.json")


TRAFFIC_FILES = ("r06_traffic.json", "r06_traffic_pca_direct1.json", "r06_traffic_cfg5_d7.json", "r06_traffic_cfg5_d1.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json")


def attach_traffic(roof, wkey, sync_launch_us, cmdline_traffic):
    """roofline.traffic = PHYSICAL HBM bytes per sweep launch of this very workload (rocprofv3 FETCH_SIZE x 2 on gfx950 + WRITE_SIZE from their own --pmc passes over
    the synchronous job, committed under profiles/: counters cannot be read inside the timed run), and what that is per second of such a launch."""
    traffic, source = cmdline_traffic, "command line" if cmdline_traffic is not None else None
    if traffic is None:
        for name in TRAFFIC_FILES:
            tp_, source = static_profile(name, wkey)
            if tp_:
                traffic = tp_["traffic_bytes_per_launch"]
                break
    roof["traffic"], roof["traffic_source"] = traffic, source if traffic is not None else None
    if traffic and sync_launch_us and sync_launch_us > 0:
        roof["physical_hbm_gbs"] = round(traffic / (sync_launch_us * 1e-6) / 1e9, 1)       # (counter passes and launch time: both of the synchronous job)
        roof["physical_hbm_frac"] = round(roof["physical_hbm_gbs"] / HBM_PEAK_GBS, 4)
    else:
        roof["physical_hbm_gbs"], roof["physical_hbm_frac"] = None, None


def valu_roofline(wkey):
    """What really bounds the sweep: vector-ALU issue (SQ counters of separate rocprofv3 --pmc passes, committed under profiles/; only a file
    taken on this very workload is used)."""
    for name in VALU_FILES:
        valu, valu_source = static_profile(name, wkey)
        if valu:
            return {"bound": "valu issue", "kernel": "k_sweep",
                    # calibrated (r04 on): the sweep's VALU-busy counter per elapsed cycle divided by the same ratio of a kernel that does
                    # nothing but issue dependent-free f32 VALU work at the same occupancy -- <= 1 by construction; older files carry the raw figure
                    "active_frac": valu.get("valu_active_frac_calibrated", valu["valu_active_frac"]),
                    "active_frac_uncalibrated": valu["valu_active_frac"], "calibration": valu.get("calibration"),
                    "wave_insts_per_64_hits": valu["valu_wave_insts_per_64_hits"],       # one wave-instruction serves 64 (point, voxel) evaluations
                    "lane_insts_per_hit": round(valu["valu_wave_insts_per_64_hits"] / 64.0, 2),
                    "physical_hbm_frac_of_peak": valu.get("physical_hbm_frac_of_peak"), "source": valu_source}
    return None


def cpu_leg(a, W, G, res_np, B):
    """(1) the reference-shaped arrangement on pair 0 at 4 threads (the nodelet's setting,
    scan_matching_odom_nodelet.cpp:110,116), 8 (launch/dlo_lfa_ggo_kitti.launch:112) and all physical cores: 3 warm-ups, then
    median / p10 / p90 over 20 repeats, target build and align separately; (2) the optimised port the same way; (3) a bounded
    batch sample with the port (total wall / count) that doubles as the pose-by-pose parity check of the GPU results."""
    from oracle import oracle_py as O
    model, phys, logical = host_info()
    kw = dict(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[a.mode], variant=1 if a.variant == "pca" else 0)
    op = O.default_params(**kw)
    tg0 = cloud_np(W, "T", 0)
    sr0 = cloud_np(W, "S", 0)
    refshape_ok = a.mode != "kdtree"

    def stats(v):
        v = np.asarray(v) * 1e3
        return {"median_ms": round(float(np.median(v)), 3), "p10_ms": round(float(np.percentile(v, 10)), 3), "p90_ms": round(float(np.percentile(v, 90)), 3)}

    def protocol(make_grid, align, threads, reps=20, warm=3):
        O.lib().ora_set_threads(threads)
        tb, ta = [], []
        for r in range(warm + reps):
            c0 = time.perf_counter()
            g = make_grid()
            c1 = time.perf_counter()
            align(g)
            c2 = time.perf_counter()
            if r >= warm:
                tb.append(c1 - c0)
                ta.append(c2 - c1)
        reg_s = 1.0 / (np.median(tb) + np.median(ta))
        return {"build": stats(tb), "align": stats(ta), "registrations_per_s": round(float(reg_s), 3)}

    quota = cpu_quota()
    usable = min(phys or logical, quota) if quota else (phys or logical)     # "all physical cores" = what the container may really use
    settings = sorted({t for t in (4, 8, usable) if t <= max(usable, 4)})
    shaped, port = {}, {}
    for th in settings:
        if refshape_ok:
            shaped[str(th)] = protocol(lambda: O.RefGrid(tg0, op), lambda g: O.ref_align(g, sr0, G), th)
        port[str(th)] = protocol(lambda: O.Grid(tg0, op), lambda g: O.align(g, sr0, G), th, reps=10, warm=2)
    best_port = max(port, key=lambda k: port[k]["registrations_per_s"])
    best_shaped = max(shaped, key=lambda k: shaped[k]["registrations_per_s"]) if shaped else None
    # (3) batch sample with the port at its best thread count; pose-by-pose parity of the GPU results (sample spread over the index range)
    parity, port_batch = parity_leg(a, W, G, res_np, B, seconds=a.cpu_seconds, threads=int(best_port))
    label = "CPU restatement of ndt_omp (reference not buildable in this environment)"
    # `value` = the FASTEST CPU arrangement measured (the optimised port at its best thread count), so that any GPU/CPU ratio taken
    # from it is the conservative one; the reference-shaped arrangement SURVEY 8(d) asks for stands next to it under its own name
    value, cores = port[best_port]["registrations_per_s"], int(best_port)
    cpu = {"value": value, "unit": "registrations/s", "cores": cores, "kind": "port",
           "arrangement_of_value": "optimised port of the oracle (dense cell table, fused transform, OpenMP over 256-point chunks)",
           "value_reference_shaped": shaped[best_shaped]["registrations_per_s"] if best_shaped is not None else None,
           "cores_reference_shaped": int(best_shaped) if best_shaped is not None else None,
           "arrangement_reference_shaped": "oracle/ndt_oracle_refshape.inc: ordered-map grid, serial build, per-point heap neighbour lists, exp(p) + dense 4x6 / 24x6 "
                                           "matrices per evaluation, cloud rewrite per sweep, schedule(guided, 8) -- SURVEY.md 8(d)" if best_shaped is not None else
                                           "not available (the reference-shaped arrangement does not cover KDTREE)",
           "sample": f"pair 0 of this workload, one registration = target build + align: 3 warm-ups then the median of 20 repeats (reference-shaped) / "
                     f"2 warm-ups then the median of 10 (port) at {settings} threads (OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}); each value = the fastest setting; {label}",
           "host": f"{model}; {phys} physical cores, {logical} logical CPUs" + (f", cgroup CPU quota {quota} CPUs (the 'all cores' setting is {usable} threads)" if quota else ""),
           "reference_shaped_by_threads": shaped, "optimised_port_by_threads": port,
           "optimised_port_batch": port_batch}
    return cpu, parity


def host_clouds_leg(a, ndt, prm, dev_index, T, S, guesses, B, N, steps, uploaders=None):
    """The drop-in path as a PCL host would drive it for a batch: every cloud starts as an array of 32-byte pcl::PointXYZI records
    in ordinary (pageable) host memory and goes through mi355ndt_batch_set_target / _set_source (staging in pinned memory, PCIe,
    AoS -> SoA on the device), then build + align.  Two engines, each driven by its own host thread with its own pool of
    uploader threads, so that one engine's uploads overlap the other's kernels (scan_matching_odom_nodelet.cpp:220-221 is the
    per-frame call pattern this batches)."""
    import threading
    B = min(B, 271)                                    # one config-3 batch at most: the leg holds every cloud twice in host memory
    T, S, guesses = T[:B], S[:B], guesses[:B]
    main_cpus_before = len(os.sched_getaffinity(0))
    if uploaders is None:                              # staging threads per engine: within the CPUs the container may use
        uploaders = max(1, min(8, ((cpu_quota() or os.cpu_count() or 8) - 4) // 2))
    rec = 8                                            # floats per PointXYZI record: x y z 1 | intensity pad pad pad
    tg = np.zeros((B, N, rec), np.float32)
    sr = np.zeros((B, N, rec), np.float32)
    tg[:, :, :3] = T.permute(0, 2, 1).cpu().numpy(); tg[:, :, 3] = 1.0
    sr[:, :, :3] = S.permute(0, 2, 1).cpu().numpy(); sr[:, :, 3] = 1.0
    stride = rec * 4
    tgp, srp = np.uint64(tg.ctypes.data), np.uint64(sr.ctypes.data)
    results = {}

    def drive(idx, nsteps, out):
        out["cpus%d" % idx] = apply_affinity()        # not the one cpu an OpenMP runtime may have bound the main thread to
        eng = ndt.Engine(prm, device=dev_index)
        eng.set_option(ndt.OPT_F32_SUM_ORDER, a.f32_sum_order)
        eng.set_option(ndt.OPT_ARITH, a.arith)
        eng.batch_reserve(B, N, N)
        res = (ndt.Result * B)()
        tptr = tgp + np.arange(B, dtype=np.uint64) * np.uint64(N * stride)
        sptr = srp + np.arange(B, dtype=np.uint64) * np.uint64(N * stride)
        cnt = np.full(B, N, np.uint64)

        phases = []

        def one():
            t0 = time.perf_counter()
            eng.batch_set_clouds_raw(0, tptr, cnt, sptr, cnt, stride, uploaders)     # the engine's own staging threads
            t1 = time.perf_counter()
            eng.batch_build_targets()
            eng.batch_align_raw(guesses, res)
            phases.append((t1 - t0, time.perf_counter() - t1))
        one()                                          # warm-up (allocations, pinned slots)
        phases.clear()
        out["barrier"].wait()
        for _ in range(nsteps):
            one()
        out["t_end%d" % idx] = time.perf_counter()  # before the teardown (freeing the pinned ring takes milliseconds)
        out["phases%d" % idx] = phases
        out[idx] = np.frombuffer(res, dtype=np.uint8).copy()
        eng.close()

    results["barrier"] = threading.Barrier(3)
    n_each = max(1, steps // 2)
    th = [threading.Thread(target=drive, args=(i, n_each, results)) for i in range(2)]
    for t in th:
        t.start()
    results["barrier"].wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = max(results.pop("t_end0"), results.pop("t_end1")) - t0
    regs = 2 * n_each * B
    ph = np.array(results.pop("phases0") + results.pop("phases1"))
    results.pop("cpus1")
    bytes_pcie = regs * 2 * N * 12                     # what crosses the link: x,y,z of both clouds
    bytes_host = regs * 2 * N * stride                 # what the staging threads read from the caller's records
    return {"registrations_per_s": round(regs / dt, 1), "ms_per_batch_of_%d" % B: round(1e3 * dt / (2 * n_each), 3),
            "pcie_h2d_gbs": round(bytes_pcie / dt / 1e9, 2), "pcie_h2d_frac_of_gen5_x16_63gbs": round(bytes_pcie / dt / 63.0e9, 3),
            "host_records_read_gbs": round(bytes_host / dt / 1e9, 2), "record_bytes": stride, "engines": 2, "cpus_of_the_driving_threads": results.pop("cpus0"), "cpus_of_the_main_thread": main_cpus_before,
            "staging_ms_per_batch_median": round(1e3 * float(np.median(ph[:, 0])), 2), "build_align_ms_per_batch_median": round(1e3 * float(np.median(ph[:, 1])), 2), "uploader_threads_per_engine": uploaders,
            "cpu_quota": cpu_quota(), "bound": "host-side staging (reading the caller's 32-byte records with the CPUs the container may use), not PCIe",
            "what": "host pcl::PointXYZI clouds (pageable memory) -> batch_set_target/_set_source -> build -> align -> results on the host"}, results



def host_clouds_stream_leg(a, ndt, prm, dev_index, T, S, guesses, B, N, steps, threads=None):
    """The drop-in path STREAMED (mi355ndt_stream_submit_host): the same host pcl::PointXYZI batches, ONE engine, three batch contexts -- batch k + 1 is staged
    (all the engine's staging threads) and crosses PCIe while the launch of batch k runs; results collected two submits later.  What a host that receives clouds
    one callback at a time (scan_matching_odom_nodelet.cpp:144-183) and batches them would call."""
    B = min(B, 271)
    T, S, guesses = T[:B], S[:B], guesses[:B]
    if threads is None:
        threads = max(2, min(14, (cpu_quota() or os.cpu_count() or 8) - 2))
    rec = 8
    tg = np.zeros((B, N, rec), np.float32)
    sr = np.zeros((B, N, rec), np.float32)
    tg[:, :, :3] = T.permute(0, 2, 1).cpu().numpy(); tg[:, :, 3] = 1.0
    sr[:, :, :3] = S.permute(0, 2, 1).cpu().numpy(); sr[:, :, 3] = 1.0
    stride = rec * 4
    tptr = np.uint64(tg.ctypes.data) + np.arange(B, dtype=np.uint64) * np.uint64(N * stride)
    sptr = np.uint64(sr.ctypes.data) + np.arange(B, dtype=np.uint64) * np.uint64(N * stride)
    cnt = np.full(B, N, np.uint64)
    apply_affinity()
    eng = ndt.Engine(prm, device=dev_index)
    eng.set_option(ndt.OPT_F32_SUM_ORDER, a.f32_sum_order)
    eng.set_option(ndt.OPT_ARITH, a.arith)
    nctx = 3
    eng.stream_begin(nctx, B, N, N)
    res = (ndt.Result * B)()
    ids, col, t_sub, t_col = [], 0, [], []

    def step():
        nonlocal col
        if len(ids) - col >= nctx:
            c0 = time.perf_counter()
            eng.stream_collect_raw(ids[col], res)
            t_col.append(time.perf_counter() - c0)
            col += 1
        c0 = time.perf_counter()
        ids.append(eng.stream_submit_host_raw(tptr, cnt, sptr, cnt, stride, guesses, threads))
        t_sub.append(time.perf_counter() - c0)
    for _ in range(nctx):                              # warm-up: allocations, pinned slots, the stream's build plan
        step()
    while col < len(ids):
        eng.stream_collect_raw(ids[col], res); col += 1
    t_sub.clear(); t_col.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    while col < len(ids):
        c0 = time.perf_counter()
        eng.stream_collect_raw(ids[col], res); col += 1
        t_col.append(time.perf_counter() - c0)
    dt = time.perf_counter() - t0
    out = np.frombuffer(res, dtype=np.uint8).copy()
    eng.stream_end()
    eng.close()
    regs = steps * B
    return {"registrations_per_s": round(regs / dt, 1), "ms_per_batch_of_%d" % B: round(1e3 * dt / steps, 3),
            "pcie_h2d_gbs": round(regs * 2 * N * 12 / dt / 1e9, 2), "pcie_h2d_frac_of_gen5_x16_63gbs": round(regs * 2 * N * 12 / dt / 63.0e9, 3),
            "host_records_read_gbs": round(regs * 2 * N * stride / dt / 1e9, 2), "record_bytes": stride, "engines": 1, "batch_contexts": nctx, "staging_threads": threads,
            "submit_ms_per_batch_median": round(1e3 * float(np.median(t_sub)), 2), "collect_wait_ms_per_batch_median": round(1e3 * float(np.median(t_col)), 2) if t_col else None,
            "cpu_quota": cpu_quota(),
            "what": "host pcl::PointXYZI clouds (pageable memory) -> mi355ndt_stream_submit_host (staging + PCIe of batch k + 1 under the launch of batch k) -> results on the host; "
                    "submit = the staging the caller waits for, collect wait = what is left of the GPU's work after it"}, out


def sequential_leg(a, ndt, dev_index, dev, n_frames, parity_frames=12):
    """Latency mode -- what the live nodelet does (scan_matching_odom_nodelet.cpp:192-261): a drive of `n_frames` scans of 65,536 points,
    every scan aligned against its keyframe with the guess carried over from the previous frame, the nodelet's own registration
    parameters (pclpca, 1.0 m, DIRECT1, eps 0.01, 64 iterations, :109-119) and keyframe thresholds (:67-76, 10 Hz stamps), through
    mi355ndt_sequence_run: host PointXYZI clouds in, guess propagation / keyframe test / target switch on the device.  Frames are
    NOT independent here (frame k's guess and target depend on frame k-1), so this is a latency number, not a throughput one."""
    from lv_slam_amd import synth
    from oracle import oracle_py as O
    scans, _ = synth.make_sequence(n_frames, a.azimuth, device=dev)
    N = a.azimuth * 64
    rec = np.zeros((n_frames, N, 8), np.float32)       # 32-byte pcl::PointXYZI records in pageable host memory
    for k, sc in enumerate(scans):
        rec[k, :, :3] = sc.cpu().numpy()
    rec[:, :, 3] = 1.0
    del scans
    stamps = [0.1 * k for k in range(n_frames)]
    prm = ndt.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES["direct1"], variant=1)
    eng = ndt.Engine(prm, device=dev_index)
    eng.set_option(ndt.OPT_ARITH, a.arith)
    frames = [rec[k] for k in range(n_frames)]
    eng.sequence_run(frames[:min(n_frames, 8)], stamps[:min(n_frames, 8)])        # warm-up: allocations, pinned slots
    best = None
    for _ in range(3):
        c0 = time.perf_counter()
        out, st = eng.sequence_run(frames, stamps)
        wall = time.perf_counter() - c0
        if best is None or wall < best[0]:
            best = (wall, out, st)
    wall, out, st = best
    its = np.array([f["iterations"] for f in out[1:]], np.float64)
    # trajectory parity on a bounded prefix against the oracle's run of the same frames (the -m gpu suite checks 65 frames)
    m = min(n_frames, parity_frames)
    par = None
    if m > 1 and a.cpu_seconds > 0:
        op = O.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES["direct1"], variant=1)
        O.lib().ora_set_threads(cpu_quota() or os.cpu_count() or 8)
        ora = O.sequence([rec[k, :, :3] for k in range(m)], stamps[:m], op)
        O.lib().ora_set_threads(0)
        outm, _ = eng.sequence_run(frames[:m], stamps[:m])
        worst, same_it, same_key = (0.0, 0.0), 0, 0
        for k in range(1, m):
            e = se3_err(ora[k]["odom"], outm[k]["odom"])
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
            same_it += int(ora[k]["iterations"] == outm[k]["iterations"])
            same_key += int(ora[k]["key_id"] == outm[k]["key_id"] and ora[k]["new_keyframe"] == outm[k]["new_keyframe"])
        par = {"frames_checked": m - 1, "max_dtrans_m": worst[0], "max_drot_rad": worst[1], "iterations_equal": same_it, "keyframe_decisions_equal": same_key,
               "what": f"odom_velo of frames 1..{m - 1} vs the oracle's run of the same frames (oracle_py.sequence)"}
    eng.close()
    aligned = n_frames - 1
    return {"frames_per_s": round(n_frames / wall, 1), "frames": n_frames, "points_per_frame": N,
            "config": "ndt_pca, 1.0 m, DIRECT1, eps 0.01, max_iter 64; keyframe thresholds 5 m / 0.17 rad / 1 s, stamps 0.1 s apart",
            "wall_ms": round(1e3 * wall, 2), "upload_ms": round(st["upload_ms"], 2), "build_all_grids_ms": round(st["build_ms"], 3), "track_ms": round(st["track_ms"], 3),
            "track_ms_per_frame": round(st["track_ms"] / max(1, aligned), 4), "aligns": aligned + (1 if n_frames > 1 else 0),
            "mean_iterations": round(float(its.mean()), 2) if len(its) else 0.0, "keyframes": int(sum(f["new_keyframe"] for f in out)),
            "update_launches": int(st["update_launches"]), "converged": int(sum(f["converged"] for f in out[1:])),
            "host_round_trips_between_frames": 0, "parity": par,
            "what": "host pcl::PointXYZI clouds -> mi355ndt_sequence_run (upload all frames, one batched build of every frame's grid, then the "
                    "per-frame loop of matching_s2k on the device: the host only pumps (update, sweep) launches)"}


def prefiltered_row(a, ndt, synth, dev, dev_index):
    """A workload conditioned the way lv_slam conditions it: the NDT input of the odometry node is /filtered_points -- the raw scan
    after PrefilteringNodelet's 0.5-100 m distance gate and 0.1 m VoxelGrid centroid down-sampling (launch/dlo_kitti.launch:30-36,
    src/lidar_odometry/prefiltering_nodelet.cpp:137-181).  Raw scans of a.azimuth * 64 points go through mi355ndt_prefilter on the
    device, the result is installed device-to-device as target / source (mi355ndt_use_prefiltered), then align.  One registration
    at a time (the prefilter entry point is the single-registration surface), latency mode on.  Not the headline: a row beside it."""
    from oracle import oracle_py as O
    n_pairs, N = a.pairs, a.azimuth * 64
    raw = []
    for k in range(n_pairs):
        t, s, _ = synth.make_pair(k, a.azimuth, device=dev)
        raw.append((np.ascontiguousarray(t.cpu().numpy()), np.ascontiguousarray(s.cpu().numpy())))
    prm = ndt.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[a.mode], variant=1 if a.variant == "pca" else 0)
    eng = ndt.Engine(prm, device=dev_index)
    eng.set_option(ndt.OPT_ARITH, a.arith)
    eng.set_latency_mode(True)
    G = synth.default_guess()
    res, t_pf, t_in, t_al, n_t, n_s, leaves, maxleaf, hpl = [], 0.0, 0.0, 0.0, [], [], [], [], []

    def one(k, timed):
        nonlocal t_pf, t_in, t_al
        tg, sr = raw[k]
        c0 = time.perf_counter()
        mt = eng.prefilter(tg, 0.5, 100.0, 0.1, fetch=False)
        c1 = time.perf_counter()
        eng.use_prefiltered(as_target=True)                        # device-to-device + voxelise (setInputTarget)
        c2 = time.perf_counter()
        ms = eng.prefilter(sr, 0.5, 100.0, 0.1, fetch=False)
        c3 = time.perf_counter()
        eng.use_prefiltered(as_target=False)
        eng.synchronize()
        c4 = time.perf_counter()
        r = eng.align(G)
        c5 = time.perf_counter()
        if timed:
            t_pf += (c1 - c0) + (c3 - c2); t_in += (c2 - c1) + (c4 - c3); t_al += c5 - c4
            res.append(r); n_t.append(mt); n_s.append(ms)
    for k in range(min(3, n_pairs)):
        one(k, False)
    eng.profile_enable(True)
    eng.profile_reset()
    gc.collect()
    gc.disable()                                  # (as in timed_job: a generation-2 collection costs ~15 ms and lands in whichever phase it interrupts)
    for k in range(n_pairs):
        one(k, True)
    gc.enable()
    prof = eng.profile_get()
    eng.profile_enable(False)
    for k in range(min(n_pairs, 8)):                            # leaf statistics of the conditioned targets
        eng.prefilter(raw[k][0], 0.5, 100.0, 0.1, fetch=False)
        eng.use_prefiltered(as_target=True)
        v = eng.get_voxels(0)
        leaves.append(len(v)); maxleaf.append(int(v["n"].max())); hpl.append(float(np.mean(v["n"][v["n"] > 0])))
    # parity against the oracle's prefilter + align on a bounded sample
    parity = None
    if a.cpu_seconds > 0:
        op = O.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[a.mode], variant=1 if a.variant == "pca" else 0)
        O.lib().ora_set_threads(cpu_quota() or os.cpu_count() or 8)
        done, t_cpu, worst, same_it, same_n = 0, 0.0, (0.0, 0.0), 0, 0
        for k in spread_order(n_pairs):
            if done >= 2 and t_cpu >= a.cpu_seconds:
                break
            c0 = time.perf_counter()
            ft, fs = O.prefilter(raw[k][0]), O.prefilter(raw[k][1])
            ro = O.align(O.Grid(ft, op), fs, G)
            t_cpu += time.perf_counter() - c0
            e = se3_err(ro["final"], res[k]["final"])
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
            same_it += int(ro["iterations"] == res[k]["iterations"])
            same_n += int(len(ft) == n_t[k] and len(fs) == n_s[k])
            done += 1
        O.lib().ora_set_threads(0)
        parity = {"pairs_checked": done, "max_dtrans_m": worst[0], "max_drot_rad": worst[1], "iterations_equal": same_it, "filtered_point_counts_equal": same_n,
                  "cpu_registrations_per_s": round(done / t_cpu, 2), "tolerance": "trans<1e-4 m, rot<1e-5 rad",
                  "oracle": "ora_prefilter + ora_align (parity unpinned, DESIGN.md 2)"}
    eng.close()
    sw_s = prof["sweep_ms"] * 1e-3
    ach = prof["sweep_alg_bytes"] / sw_s / 1e9 if sw_s > 0 else 0.0
    tot = t_pf + t_in + t_al
    out = {"metric": "NDT registrations/sec (64k-pt Velodyne pairs)", "value": round(n_pairs / tot, 2), "unit": "registrations/s", "n_gpus": 1,
           "steps": 1, "warmup": 1, "ms_per_step": round(1e3 * tot, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 terms, f64 accumulation", "data": "synthetic",
           "config": {"workload": f"reference-conditioned row: {n_pairs} synthetic HDL-64E scan pairs of {N} raw points each -> device prefilter (0.5-100 m gate, 0.1 m VoxelGrid) -> "
                                  f"ndt_{a.variant}, {a.resolution} m voxels, {a.mode.upper()}, eps 0.01, max_iter 64; ONE registration at a time (host raw clouds in, latency mode)",
                      "pairs_total": n_pairs, "raw_points_per_cloud": N, "filtered_points_per_target_mean": round(float(np.mean(n_t)), 1),
                      "filtered_points_per_source_mean": round(float(np.mean(n_s)), 1), "searchable_leaves_per_target_mean": round(float(np.mean(leaves)), 1),
                      "points_per_leaf_mean": round(float(np.mean(hpl)), 2), "largest_leaf": int(max(maxleaf)),
                      "mean_iterations": round(float(np.mean([r["iterations"] for r in res])), 2), "converged": int(sum(r["converged"] for r in res))},
           "registrations_per_s_without_prefilter_time": round(n_pairs / (t_in + t_al), 2),
           "ms_per_registration": {"prefilter_x2_incl_upload": round(1e3 * t_pf / n_pairs, 4), "install_target_and_source_incl_voxelise": round(1e3 * t_in / n_pairs, 4),
                                   "align": round(1e3 * t_al / n_pairs, 4)},
           "roofline": {"bound": "hbm", "kernel": "k_sweep (latency mode)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None, "launches": prof["sweep_launches"], "avg_launch_us": round(1e3 * prof["sweep_ms"] / max(1, prof["sweep_launches"]), 2),
                        "hits_per_point": round(prof["sweep_hits"] / max(1, prof["sweep_points"]), 3),
                        "note": "one pair per launch: the GPU is mostly idle by construction; the batched headline is the throughput number"},
           "cpu_baseline": None, "parity": parity}
    print(json.dumps(out), flush=True)

RES_DT = np.dtype([("final", "<f4", 16), ("tp", "<f8"), ("score", "<f8"), ("it", "<i4"), ("conv", "<i4"), ("sweeps", "<i4"), ("status", "<i4"), ("hits", "<i8")])


def spread_order(n):
    """Indices 0..n-1 in an order whose every prefix covers the whole range evenly (bit-reversal / van der Corput), with the last
    index moved to second place: a time-bounded parity sample then spans the job's index range instead of its first pairs."""
    bits = max(1, (n - 1).bit_length())
    order = sorted(range(n), key=lambda i: int(format(i, f"0{bits}b")[::-1], 2))
    if n > 2:
        order.remove(n - 1)
        order.insert(1, n - 1)
    return order


def parity_leg(a, W, G, res_np, B, seconds, threads):
    """Pose-by-pose check of the GPU results of one job against the oracle on a time-bounded sample that is spread over the whole
    index range of the job (spread_order); doubles as the batch rate of the optimised port."""
    from oracle import oracle_py as O
    kw = dict(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[a.mode], variant=1 if a.variant == "pca" else 0)
    op = O.default_params(**kw)
    O.lib().ora_set_threads(int(threads))
    O.lib().ora_set_variant(1 if getattr(a, "f32_sum_order", 0) == 1 else 0, 256)     # the oracle's matching f32 sum order (ORA_VAR_SUM3_02_1)
    done, t_cpu, worst, it_match, conv_match, idx_seen, beyond = 0, 0.0, (0.0, 0.0), 0, 0, [], 0
    for k in spread_order(B):
        if done >= 3 and t_cpu >= seconds:
            break
        tg = cloud_np(W, "T", k)
        sr = cloud_np(W, "S", k)
        c0 = time.perf_counter()
        ro = O.align(O.Grid(tg, op), sr, G)
        t_cpu += time.perf_counter() - c0
        e = se3_err(ro["final"], res_np["final"][k].reshape(4, 4).T)
        worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        beyond += int(e[0] >= 1e-4 or e[1] >= 1e-5)
        it_match += int(ro["iterations"] == int(res_np["it"][k]))
        conv_match += int(bool(ro["converged"]) == bool(res_np["conv"][k]))
        idx_seen.append(k)
        done += 1
    O.lib().ora_set_threads(0)
    O.lib().ora_set_variant(0, 256)
    parity = {"pairs_checked": done, "max_dtrans_m": worst[0], "max_drot_rad": worst[1], "iterations_equal": it_match, "converged_flags_equal": conv_match,
              "pairs_beyond_tolerance": beyond, "f32_sum_order": getattr(a, "f32_sum_order", 0), "arith": getattr(a, "arith", 0),
              "sample": f"{done} of the {B} pairs of rank 0, spread over the whole index range (bit-reversal order; slots {min(idx_seen)}..{max(idx_seen)} touched)",
              "tolerance": "trans<1e-4 m, rot<1e-5 rad", "oracle": "parity unpinned (no reference-originated vectors exist, DESIGN.md 2; error bar: BASELINE.md 5)",
              "note": "pairs that never converge (iterations = max_iterations + 2, e.g. ndt_pca with DIRECT26 where the compounding "
                      "weights make the iteration oscillate) amplify rounding-order differences and are not comparable pose by pose"}
    return parity, {"registrations_per_s": round(done / max(t_cpu, 1e-9), 3), "threads": int(threads), "sample": f"{done} of the {B} pairs, total wall / count"}


class Ctx:
    """What every leg of a run shares: the rank's place in the job, its device, the process group (or None)."""
    rank = 0; world = 1; local = 0; dev = None; dist = None; backend = "nccl"; use_dist = False; on_dev = True; shard = None; ndt = None; G = None; stream_reserve = None


def cloud_np(W, side, k):
    """Pair slot k's target ("T") or source ("S") cloud of workload W as an [n, 3] f32 host array (ragged clouds: their own count)."""
    n = int(W["tcnt" if side == "T" else "scnt"][k])
    return W[side][k, :, :n].T.contiguous().cpu().numpy()


def generate_synthetic(ctx, synth, ids, azimuth):
    """This rank's scan pairs, ray-cast on ITS GPU and left resident in HBM as [pair][3][N] SoA rows.  The CPU only draws every pair's
    seeded noise (torch's CPU generator releases the GIL: a few Python threads, each inside the CPUs the container may use) and
    looks up the cached street primitives; eight ranks of a node therefore do not queue on the host's CPU quota."""
    N = azimuth * 64
    t0 = time.perf_counter()
    apply_affinity()                                             # (the thread that issues the device work: not on the one CPU libgomp bound it to)
    T = torch.empty(len(ids), 3, N, device=ctx.dev, dtype=torch.float32)
    S = torch.empty(len(ids), 3, N, device=ctx.dev, dtype=torch.float32)
    quota = cpu_quota() or os.cpu_count() or 8
    # torch sizes its CPU thread pool to the CPUs it can see; under a cgroup quota (16 of 256 CPUs on the GPU boxes) that many threads
    # only throttle each other: keep torch's pool within the quota
    torch.set_num_threads(max(1, min(torch.get_num_threads(), quota // 2)))

    # The seeded draws of a pair (2 x N f64 normals + 2 x N uniforms from torch's CPU generator, GIL released) come from four threads (more
    # only take the CPU from the thread that issues the device work: 12 threads 9.0 s, 4 threads 6.5 s for 4,541 pairs), three batches ahead
    # of the device, straight into pinned staging buffers (one asynchronous copy per batch); the ray casting of a batch of pairs
    # is ONE pass of batched device operations (synth.make_pairs: the same arithmetic as synth.make_pair scan by scan, bit for bit --
    # tests/test_synth.py -- without ~150 kernel launches from Python per pair, which is what the 29 s of rounds 3-5 were).
    per = max(1, min(16, (16 * 65536) // N))
    ahead = int(os.environ.get("BENCH_GEN_AHEAD", "3"))
    on_gpu = ctx.dev.type == "cuda"
    ring_bufs = [(torch.empty(per, 2, N, dtype=torch.float64, pin_memory=on_gpu), torch.empty(per, 2, N, dtype=torch.float64, pin_memory=on_gpu))
                 for _ in range(ahead + 1)]
    copied = [None] * (ahead + 1)                                # the event behind the last copy out of each staging slot

    def draw(k, slot, j):
        apply_affinity()
        synth.pair_noise(ids[k], N, noise_sigma=None, out=(ring_bufs[slot][0][j], ring_bufs[slot][1][j]))

    from concurrent.futures import ThreadPoolExecutor
    batches = [list(range(b, min(b + per, len(ids)))) for b in range(0, len(ids), per)]
    with ThreadPoolExecutor(max_workers=int(os.environ.get("BENCH_GEN_THREADS", max(2, min(4, quota // 4))))) as pool:
        futs, waited = {}, 0.0

        def submit(bi):
            slot = bi % (ahead + 1)
            if copied[slot] is not None:
                copied[slot].synchronize()
            futs[bi] = [pool.submit(draw, k, slot, j) for j, k in enumerate(batches[bi])]
        for bi in range(min(ahead, len(batches))):
            submit(bi)
        for bi, ks in enumerate(batches):
            if bi + ahead < len(batches):
                submit(bi + ahead)
            tw = time.perf_counter()
            for f in futs.pop(bi):
                f.result()
            waited += time.perf_counter() - tw
            slot = bi % (ahead + 1)
            noise = ring_bufs[slot][0][:len(ks)].to(ctx.dev, non_blocking=True)
            ring = ring_bufs[slot][1][:len(ks)].to(ctx.dev, non_blocking=True)
            if on_gpu:
                copied[slot] = torch.cuda.Event()
                copied[slot].record()
            t, s_, _ = synth.make_pairs([ids[k] for k in ks], azimuth, device=ctx.dev, draws=(noise, ring), unit_noise=True)
            T[ks[0]:ks[-1] + 1] = t.permute(0, 2, 1)
            S[ks[0]:ks[-1] + 1] = s_.permute(0, 2, 1)
            del t, s_, noise, ring
    if on_gpu:
        torch.cuda.synchronize()
    if os.environ.get("BENCH_GEN_DEBUG"):
        print(f"[generate_synthetic] {len(ids)} pairs: {time.perf_counter() - t0:.2f} s, of which waiting for the draws {waited:.2f} s "
              f"({per} pairs per pass, {ahead} batches ahead)", file=sys.stderr)
    cnt = [N] * len(ids)
    return {"T": T, "S": S, "tcnt": cnt, "scnt": list(cnt), "pitch": N, "ids": list(ids), "data": "synthetic", "max_points": N,
            "generated_on": f"cuda:{ctx.local} (ray casting in torch on the device, {per} pairs per pass; per-pair noise drawn on the CPU)"}, time.perf_counter() - t0


def load_kitti(ctx, a, ndt, ids):
    """--kitti-dir: pair k = (target frame k, source frame k + 1) of a KITTI odometry sequence (scripts/lidar_odom_kitti.sh:6 plays the
    same files), optionally through the device prefilter the reference's launch file puts in front of the odometry node
    (launch/dlo_kitti.launch:30-36).  Ragged point counts: rows padded to the longest cloud, the counts travel with the batch."""
    from lv_slam_amd import kitti
    t0 = time.perf_counter()
    files = kitti.list_frames(a.kitti_dir)
    need = max(ids) + 2
    if len(files) < need:
        raise SystemExit(f"--kitti-dir {a.kitti_dir}: {len(files)} frames, the job needs {need}")
    pf = ndt.Engine(ndt.default_params(), device=ctx.local) if a.kitti_prefilter else None
    cache = {}

    def frame(f):
        if f not in cache:
            xyz = kitti.load_frame(files[f])
            cache[f] = pf.prefilter(xyz, 0.5, 100.0, 0.1, fetch=True) if pf is not None else xyz
            for old in [q for q in cache if q < f - 1]:
                del cache[old]
        return cache[f]
    clouds = [(frame(k), frame(k + 1)) for k in ids]
    if pf is not None:
        pf.close()
    T, S, tcnt, scnt, pitch = kitti.pack_soa(clouds, ctx.dev)
    return {"T": T, "S": S, "tcnt": tcnt, "scnt": scnt, "pitch": pitch, "ids": list(ids), "data": "kitti", "max_points": max(max(tcnt), max(scnt)),
            "generated_on": f"{os.path.abspath(a.kitti_dir)} ({len(files)} frames)" + (", device prefilter 0.5-100 m + 0.1 m VoxelGrid" if pf is not None else "")}, time.perf_counter() - t0


def timed_job(ctx, eng, W, nb, job_total, steps, warmup, min_seconds=0.5):
    """`warmup` untimed + `steps` timed passes over the first `nb` pair slots of workload W (this rank's shard of a job of `job_total`
    pairs): voxelise every target, align every pair, and -- when a process group exists -- pack the pose records on the device and
    all-gather them.  Barrier + synchronize on both sides of the timed steps, MAX over ranks.  steps = None: as many as fill
    `min_seconds`.  Returns timings, results, profile, gather check."""
    ndt, shard, dist, dev, on_dev, use_dist = ctx.ndt, ctx.shard, ctx.dist, ctx.dev, ctx.on_dev, ctx.use_dist
    ids = W["ids"][:nb]
    cap = shard.shard_capacity(job_total, ctx.world)
    eng.batch_bind_device(W["T"].data_ptr(), W["tcnt"][:nb], W["pitch"], W["S"].data_ptr(), W["scnt"][:nb], W["pitch"])
    guesses = np.ascontiguousarray(np.broadcast_to(ctx.G.T.reshape(1, 16), (nb, 16)), dtype=np.float32)
    res = (ndt.Result * nb)()
    res_np = np.frombuffer(res, dtype=RES_DT)
    # pose records: packed by the engine on the device into this tensor, which goes straight into the all-gather
    rec_dev = torch.empty(cap, shard.REC_WORDS, device=dev, dtype=torch.int32)
    rec_host = None if on_dev else torch.empty(cap, shard.REC_WORDS, dtype=torch.int32).pin_memory()   # gloo functional check only
    gathered = torch.empty(ctx.world * cap, shard.REC_WORDS, device=dev if on_dev else "cpu", dtype=torch.int32) if use_dist else None
    gather_ev, gather_host_s = [], [0.0]

    def step(timed=False):
        eng.batch_build_targets()                 # setInputTarget for every pair: voxelise
        eng.batch_align_raw(guesses, res)         # align every pair (synchronous: results on the host)
        if use_dist:                              # pose gather: 96 B per pair, no host hop on the RCCL path
            if gather_ev:
                gather_ev[-1][1].synchronize()    # the previous gather has read rec_dev before it is packed again
            eng.batch_pose_records(ctx.rank, ctx.world, rec_dev.data_ptr(), cap)
            h0 = time.perf_counter()
            if on_dev:                            # HIP events on torch's current stream, which the collective is ordered on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                shard.gather_records(rec_dev, gathered)
                e1.record()
                if timed:
                    gather_ev.append((e0, e1))
            else:
                rec_host.copy_(rec_dev)
                shard.gather_records(rec_host, gathered)
                if timed:
                    gather_host_s[0] += time.perf_counter() - h0

    eng.profile_enable(True)                      # the warm-up runs exactly what the timed steps run (event pool touched, too)
    gc.collect()
    gc.disable()                                  # a generation-2 collection (torch + numpy object graphs) costs ~15 ms: keep it out of
    for _ in range(warmup):                       # the timed loop -- and out of the gap before it, where an idle GPU drops its clocks
        step()
    if steps is None:                             # no --steps: size the timed region (the driver passes --steps and is obeyed)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - c0, 1e-4)
        st = torch.tensor([float(max(20 if min_seconds >= 0.5 else 5, int(np.ceil(min_seconds / one))))], dtype=torch.float64, device=dev if on_dev else "cpu")
        if dist is not None:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        steps = int(st.item())
    eng.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms, tp = [], t0
    for _ in range(steps):
        step(True)                                # synchronous: batch_align returns with the results on the host
        tn = time.perf_counter()
        step_ms.append(1e3 * (tn - tp))
        tp = tn
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    prof = eng.profile_get()
    eng.profile_enable(False)
    gather_ms = None
    if use_dist:
        gather_ms = (sum(e0.elapsed_time(e1) for e0, e1 in gather_ev) if on_dev else 1e3 * gather_host_s[0]) / max(1, steps)
    gather_check = None
    if dist is not None:
        tt = torch.tensor([dt, gather_ms], device=dev if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, gather_ms = float(tt[0].item()), float(tt[1].item())
        # gather check: pair ids form a permutation of the global index space, and my own records came back bit-identical
        got = shard.unpack_records(gathered)
        perm = sorted(got) == list(range(job_total))
        same = all(np.array_equal(got[pid]["final"], res_np["final"][k].reshape(4, 4).T) and got[pid]["iterations"] == int(res_np["it"][k])
                   and got[pid]["converged"] == bool(res_np["conv"][k]) and np.float32(res_np["score"][k]) == np.float32(got[pid]["score"])
                   for k, pid in enumerate(ids))
        assert perm, "pose gather lost or duplicated pairs"
        assert same, "gathered records differ from this rank's results"
        ok = torch.tensor([int(perm and same)], dtype=torch.int32, device=dev if on_dev else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        gather_check = {"pairs_gathered": len(got), "permutation_of_all_pair_ids": bool(perm), "own_records_bit_identical_on_every_rank": bool(ok.item()),
                        "record_bytes": 96, "records_per_rank": cap, "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                        "packed_on_device": True, "host_hop": not on_dev,
                        "timed_with": "HIP events around all_gather_into_tensor on the stream the collective is ordered on, max over ranks" if on_dev else
                                      "host clock around the device-to-host copy + gloo all_gather (functional check only), max over ranks"}
    return {"dt": dt, "steps": steps, "step_ms": step_ms, "prof": prof, "res": res, "res_np": res_np, "guesses": guesses,
            "gather_ms_per_step": None if gather_ms is None else round(gather_ms, 4), "gather_check": gather_check, "B": nb}


def timed_stream_job(ctx, eng, W, nb, job_total, steps, warmup, n_batches=3, n_contexts=3, ref=None):
    """The same job as timed_job, STREAMED (mi355ndt_stream_*): step i is batch i mod `n_batches` -- distinct batches of `nb` pair slots of W --
    submitted without waiting for the one before; a launch hands its last unfinished pairs to the next, results are collected `n_contexts`
    steps later (scan_matching_odom_nodelet.cpp:144-183: the node consumes a stream of frames; BASELINE config 3: "streamed through 1 GPU").
    Timed like timed_job: barrier + synchronize on both sides of EXACTLY `steps` steps, every step's results on the host inside the region
    (the last ones through the flush).  `ref`: per distinct batch the synchronous results, compared word for word after the clock stops."""
    ndt, shard, dist, dev, on_dev, use_dist = ctx.ndt, ctx.shard, ctx.dist, ctx.dev, ctx.on_dev, ctx.use_dist
    avail = len(W["ids"]) // nb
    n_batches = max(1, min(n_batches, avail))
    cap = shard.shard_capacity(job_total, ctx.world)
    fsz = 4 * 3 * W["pitch"]                                  # bytes per pair slot of a cloud buffer
    slots = [(W["T"].data_ptr() + k * nb * fsz, W["tcnt"][k * nb:(k + 1) * nb], W["S"].data_ptr() + k * nb * fsz, W["scnt"][k * nb:(k + 1) * nb]) for k in range(n_batches)]
    guesses = np.ascontiguousarray(np.broadcast_to(ctx.G.T.reshape(1, 16), (nb, 16)), dtype=np.float32)
    res = [(ndt.Result * nb)() for _ in range(n_batches)]
    # pose records: one device block per resident batch, filled by the device as that batch's pairs finish (mi355ndt_stream_pose_records);
    # a collected batch's block goes straight into the all-gather -- no packing kernel, no host hop on the RCCL path
    rec_dev = [torch.empty(cap, shard.REC_WORDS, device=dev, dtype=torch.int32) for _ in range(n_contexts)] if use_dist else None
    rec_host = None if (on_dev or not use_dist) else torch.empty(cap, shard.REC_WORDS, dtype=torch.int32).pin_memory()   # gloo functional check only
    gathered = torch.empty(ctx.world * cap, shard.REC_WORDS, device=dev if on_dev else "cpu", dtype=torch.int32) if use_dist else None
    gather_ev, gather_host_s = [], [0.0]
    if ctx.stream_reserve is not None:
        eng.set_option(ndt.OPT_STREAM_RESERVE, ctx.stream_reserve)
    eng.stream_begin(n_contexts, nb, W["pitch"], W["pitch"])
    state = {"sub": 0, "col": 0, "ids": []}

    def gather(i, timed):
        rd = rec_dev[i % n_contexts]                   # the block batch i was submitted with
        h0 = time.perf_counter()
        if on_dev:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            shard.gather_records(rd, gathered)
            e1.record()
            state["last_gather"] = e1
            if timed:
                gather_ev.append((e0, e1))
        else:
            rec_host.copy_(rd)
            shard.gather_records(rec_host, gathered)
            if timed:
                gather_host_s[0] += time.perf_counter() - h0

    def collect_one(timed):
        i = state["col"]
        eng.stream_collect_raw(state["ids"][i], res[i % n_batches])
        state["col"] += 1
        if use_dist:
            gather(i, timed)

    def step(timed=False):
        if state["sub"] - state["col"] >= n_contexts:
            collect_one(timed)
        k = state["sub"] % n_batches
        T, tc, S, sc = slots[k]
        if use_dist:
            if on_dev and state.get("last_gather") is not None:
                state["last_gather"].synchronize()     # (a gather may still be reading the block this batch is about to reuse)
            eng.stream_pose_records(rec_dev[state["sub"] % n_contexts].data_ptr(), cap, ctx.rank, ctx.world)
        state["ids"].append(eng.stream_submit(T, tc, W["pitch"], S, sc, W["pitch"], guesses))
        state["sub"] += 1

    def drain(timed=False):
        while state["col"] < state["sub"]:
            collect_one(timed)

    eng.profile_enable(True)
    gc.collect()
    gc.disable()
    for _ in range(max(warmup, n_contexts)):           # (the first submits build synchronously: they make the stream's build plan)
        step()
    drain()
    eng.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms, tp = [], t0
    for _ in range(steps):
        step(True)
        tn = time.perf_counter()
        step_ms.append(1e3 * (tn - tp))
        tp = tn
    drain(True)                                        # every step's results are on the host before the clock stops
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    prof = eng.profile_get()
    eng.profile_enable(False)
    eng.stream_end()
    gather_ms = None
    if use_dist:
        gather_ms = (sum(e0.elapsed_time(e1) for e0, e1 in gather_ev) if on_dev else 1e3 * gather_host_s[0]) / max(1, steps)
    if dist is not None:
        tt = torch.tensor([dt, gather_ms], device=dev if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, gather_ms = float(tt[0].item()), float(tt[1].item())
    gather_check = None
    if dist is not None:
        k_last = (state["col"] - 1) % n_batches
        got = shard.unpack_records(gathered)
        mine = np.frombuffer(res[k_last], dtype=RES_DT)
        perm = sorted(got) == list(range(job_total))
        own = all(np.array_equal(got[ctx.rank + j * ctx.world]["final"], mine["final"][j].reshape(4, 4).T) and got[ctx.rank + j * ctx.world]["iterations"] == int(mine["it"][j])
                  for j in range(nb)) if perm else False
        assert perm and own, "pose gather of the streamed job lost, duplicated or changed records"
        ok = torch.tensor([int(perm and own)], dtype=torch.int32, device=dev if on_dev else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        gather_check = {"pairs_gathered": len(got), "permutation_of_all_pair_ids": bool(perm), "own_records_bit_identical_on_every_rank": bool(ok.item()),
                        "record_bytes": 96, "records_per_rank": cap, "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                        "packed_on_device": True, "host_hop": not on_dev,
                        "packed_by": "the wave that finalises a pair writes its 96-byte record into the batch's gather block (mi355ndt_stream_pose_records): no packing kernel",
                        "timed_with": "HIP events around all_gather_into_tensor on the stream the collective is ordered on, max over ranks" if on_dev else
                                      "host clock around the device-to-host copy + gloo all_gather (functional check only), max over ranks"}
    same = None
    if ref is not None:                                # word for word against the synchronous align of the same batches
        same = all(bytes(res[k]) == bytes(ref[k]) for k in range(min(n_batches, len(ref))))
        assert same, "streamed results differ from the synchronous results of the same pairs"
    return {"dt": dt, "steps": steps, "step_ms": step_ms, "prof": prof, "res": res[0], "res_np": np.frombuffer(res[0], dtype=RES_DT), "guesses": guesses,
            "gather_ms_per_step": None if gather_ms is None else round(gather_ms, 4), "gather_check": gather_check, "B": nb, "n_batches": n_batches, "n_contexts": n_contexts,
            "bit_identical_to_synchronous": same,
            "launches": int(prof["stream_launches"]), "pairs_handed_over": int(prof["stream_carried"]), "batches_rerun": int(prof["stream_redone"]),
            "launches_that_gave_up": int(prof["async_fallbacks"])}


def sync_reference(ctx, eng, W, nb, n_batches):
    """synchronous results (build + batch_align) of the first `n_batches` distinct batches of `nb` pair slots: what the stream must reproduce"""
    ndt = ctx.ndt
    n_batches = max(1, min(n_batches, len(W["ids"]) // nb))
    fsz = 4 * 3 * W["pitch"]
    guesses = np.ascontiguousarray(np.broadcast_to(ctx.G.T.reshape(1, 16), (nb, 16)), dtype=np.float32)
    out = []
    for k in range(n_batches):
        eng.batch_bind_device(W["T"].data_ptr() + k * nb * fsz, W["tcnt"][k * nb:(k + 1) * nb], W["pitch"], W["S"].data_ptr() + k * nb * fsz, W["scnt"][k * nb:(k + 1) * nb], W["pitch"])
        eng.batch_build_targets()
        r = (ndt.Result * nb)()
        eng.batch_align_raw(guesses, r)
        out.append(r)
    return out


F32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: vector f32 peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz, x2 for packed)
FLOP_PER_HIT = 450.0          # ~450 flop per (point, voxel) evaluation


def sweep_roofline(J):
    prof, dt, steps = J["prof"], J["dt"], J["steps"]
    sw_s = prof["sweep_ms"] * 1e-3
    ach = (prof["sweep_alg_bytes"] / sw_s / 1e9) if sw_s > 0 else 0.0
    tfl = (FLOP_PER_HIT * prof["sweep_hits"] / sw_s / 1e12) if sw_s > 0 else 0.0
    one_launch = prof["update_launches"] == 0 and (prof["sweep_launches"] == steps or prof.get("stream_launches", 0) > 0)   # ndt_async.hpp: the whole batch align is one launch
    return {"bound": "hbm", "kernel": "k_align_async (one launch per batch align: every derivative sweep and Newton update of every pair)" if one_launch else "k_sweep",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "frac_of_achievable_6290": round(ach / 6290.0, 4),     # MI355X_MICROARCH.md: measured-achievable HBM rate
            # stream mode with the next batch's build beside the launch (MI355NDT_OPT_STREAM_RESERVE): the launch runs on launch_slots of the GPU's
            # launch_slots + reserved_slots workgroup slots and shares caches and HBM with the build for as long as that runs -- its duration, hence `frac`,
            # includes what it gave away; `frac_of_its_slots` = frac / (launch_slots / all slots); `roofline_synchronous` is the launch with the GPU to itself
            **({"launch_slots": int(prof["stream_launch_slots"]), "reserved_slots_for_the_next_build": int(prof["stream_reserved_slots"]),
                "frac_of_its_slots": round(ach / HBM_PEAK_GBS * (prof["stream_launch_slots"] + prof["stream_reserved_slots"]) / prof["stream_launch_slots"], 4)}
               if prof.get("stream_reserved_slots", 0) > 0 else {}),
            "launches": prof["sweep_launches"], "avg_launch_us": round(1e3 * prof["sweep_ms"] / max(1, prof["sweep_launches"]), 2),
            "alg_bytes_per_launch": round(prof["sweep_alg_bytes"] / max(1, prof["sweep_launches"])),
            "hits_per_point": round(prof["sweep_hits"] / max(1, prof["sweep_points"]), 3),
            # the secondary figure of SURVEY 8(d): ~450 flop per (point, voxel) evaluation against the vector f32 peak (no MFMA on this path)
            "flops": {"achieved_tflops": round(tfl, 2), "peak_tflops": F32_PEAK_TFLOPS, "frac": round(tfl / F32_PEAK_TFLOPS, 4), "flop_per_hit": FLOP_PER_HIT,
                      "hits": int(prof["sweep_hits"])},
            "sweep_share_of_step": round(sw_s / dt, 3),
            "build_ms_per_step": round(prof["build_ms"] / max(1, steps), 3),
            "update_ms_per_step": round(prof["update_ms"] / max(1, steps), 3),
            "sweep_ms_per_step": round(prof["sweep_ms"] / max(1, steps), 3),
            "step_ms_min_median_max": [round(min(J["step_ms"]), 3), round(float(np.median(J["step_ms"])), 3), round(max(J["step_ms"]), 3)],
            "build_achieved_gbs": round(prof["build_alg_bytes"] / max(1e-9, prof["build_ms"] * 1e-3) / 1e9, 1),
            "build_frac": round(prof["build_alg_bytes"] / max(1e-9, prof["build_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def pose_deltas(res_a, res_b):
    """pair by pair: SE(3) distance of the final poses, iteration / converged flips, pairs beyond north_star's tolerance"""
    n = len(res_a)
    d = np.array([se3_err(res_a["final"][k].reshape(4, 4).T, res_b["final"][k].reshape(4, 4).T) for k in range(n)], dtype=np.float64).reshape(n, 2)
    flips = res_a["it"] != res_b["it"]
    beyond = (d[:, 0] >= 1e-4) | (d[:, 1] >= 1e-5)
    flagged = res_b["status"] == 1                 # MI355NDT_WARN_TOLERANCE_ARITH: the engine's own caveat (few hits / iteration cap); the drop-in re-runs those in the default arithmetic
    return {"pairs": n, "iteration_flips": int(flips.sum()), "converged_flag_flips": int((res_a["conv"] != res_b["conv"]).sum()),
            "pairs_beyond_tolerance": int(beyond.sum()), "pairs_beyond_tolerance_slots": [int(k) for k in np.nonzero(beyond)[0][:16]],
            "pairs_flagged_by_the_engine": int(flagged.sum()), "pairs_beyond_tolerance_and_not_flagged": int((beyond & ~flagged).sum()),
            "max_dtrans_m": float(d[:, 0].max()), "max_drot_rad": float(d[:, 1].max()), "median_dtrans_m": float(np.median(d[:, 0])),
            "p99_dtrans_m": float(np.percentile(d[:, 0], 99)), "tolerance": "trans<1e-4 m, rot<1e-5 rad"}


TOLERANCE_WHAT = ("the same job under MI355NDT_OPT_ARITH = 1 (tolerance arithmetic: fused multiply-adds, hardware exp2, symmetric inverse covariance, 37 f32 sums per lane and "
                  "work item widened to f64 from the row on, tree leaf sums in the target build; the point transform and the voxel lookup untouched) -- held to north_star's "
                  "SE(3) tolerance, not to the oracle's bits; never `value`")


def tolerance_block(ctx, a, prm, W, nb, job_total, steps, warmup, exact_res_np, want_stream, oracle_seconds, min_seconds=0.5, b=None):
    """One job again in the tolerance arithmetic: timed like the exact job (synchronous and, where the exact job was, streamed), every pair's result compared with the
    exact arithmetic's result of the same pair (which equals the oracle's wherever it was checked), and a bounded oracle sample of its own."""
    ndt = ctx.ndt
    eng = ndt.Engine(prm, device=ctx.local)
    eng.set_option(ndt.OPT_ARITH, 1)
    J = timed_job(ctx, eng, W, nb, job_total, steps, warmup, min_seconds=min_seconds)
    r_sync = sweep_roofline(J)
    res_np = np.frombuffer(np.frombuffer(J["res"], dtype=np.uint8).copy(), dtype=RES_DT)
    JS = None
    if want_stream and len(W["ids"]) >= 2 * nb:
        ref_sync = sync_reference(ctx, eng, W, nb, a.stream_batches)
        JS = timed_stream_job(ctx, eng, W, nb, job_total, J["steps"], max(2, warmup), a.stream_batches, a.stream_contexts, ref_sync)
    r_stream = sweep_roofline(JS) if JS is not None else None
    eng.close()
    out = {"what": TOLERANCE_WHAT,
           "value_tolerance_mode": round(job_total * J["steps"] / (JS if JS is not None else J)["dt"], 2), "value_mode": "streamed" if JS is not None else "synchronous",
           "value_tolerance_mode_synchronous": round(job_total * J["steps"] / J["dt"], 2), "ms_per_step_synchronous": round(1e3 * J["dt"] / J["steps"], 3),
           "value_tolerance_mode_streamed": None if JS is None else round(job_total * J["steps"] / JS["dt"], 2), "ms_per_step_streamed": None if JS is None else round(1e3 * JS["dt"] / J["steps"], 3),
           "steps": J["steps"],
           "stream": None if JS is None else {k: JS[k] for k in ("n_batches", "n_contexts", "bit_identical_to_synchronous", "launches", "pairs_handed_over", "batches_rerun", "launches_that_gave_up")},
           "roofline_synchronous": {k: r_sync[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "alg_bytes_per_launch", "build_ms_per_step", "sweep_ms_per_step", "build_frac")},
           "roofline_streamed": None if JS is None else {k: r_stream[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "alg_bytes_per_launch", "build_ms_per_step", "sweep_ms_per_step")},
           "mean_iterations": round(float(res_np["it"].mean()), 3),
           "vs_exact_arithmetic": None if exact_res_np is None else pose_deltas(exact_res_np[:nb], res_np[:nb])}
    if oracle_seconds > 0:
        import argparse
        bb = argparse.Namespace(**{**vars(b if b is not None else a), "arith": 1})
        out["parity_vs_oracle"], _ = parity_leg(bb, W, ctx.G, res_np, nb, seconds=oracle_seconds, threads=cpu_quota() or os.cpu_count() or 8)
    return out


def other_configs_block(ctx, a, synth, W_head):
    """The BASELINE configurations the headline does not time, each as its own small job on this GPU (own engine, >= --other-seconds of
    timed steps, HIP-event roofline of its sweep, oracle parity on a bounded sample): the live nodelet's registration (ndt_pca, DIRECT1,
    1 m: scan_matching_odom_nodelet.cpp:109-119) on the headline's clouds, and BASELINE config 5's per-GPU share (ndt_pca, 0.5 m, clouds
    of twice the points) with DIRECT7 and with DIRECT1.  Sizes follow --pairs / --azimuth (271 x 65,536 and 128 x 131,072 by default)."""
    import argparse
    ndt = ctx.ndt
    p5, az5 = max(1, a.pairs * 128 // 271), 2 * a.azimuth
    t0 = time.perf_counter()
    nb5 = max(1, a.stream_batches) if not a.no_stream else 1
    W5, gen5 = generate_synthetic(ctx, synth, list(range(p5 * nb5)), az5)
    specs = [("ndt_pca_direct1", dict(variant="pca", mode="direct1", resolution=1.0), W_head, min(a.pairs, len(W_head["ids"])), a.azimuth),   # (W_head holds the headline's distinct batches)
             ("config5_direct7", dict(variant="pca", mode="direct7", resolution=0.5), W5, p5, az5),
             ("config5_direct1", dict(variant="pca", mode="direct1", resolution=0.5), W5, p5, az5)]
    out = {}
    for name, kw, W, nb, az in specs:
        b = argparse.Namespace(**{**vars(a), **kw, "pairs": nb, "azimuth": az})
        prm = ndt.default_params(resolution=b.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[b.mode], variant=1)
        eng = ndt.Engine(prm, device=ctx.local)
        eng.set_option(ndt.OPT_F32_SUM_ORDER, a.f32_sum_order)
        eng.set_option(ndt.OPT_ARITH, a.arith)
        J = timed_job(ctx, eng, W, nb, nb, None, 2, min_seconds=a.other_seconds)
        r_sync = sweep_roofline(J)
        res_np = np.frombuffer(np.frombuffer(J["res"], dtype=np.uint8).copy(), dtype=RES_DT)
        JS = None
        if not a.no_stream and len(W["ids"]) >= 2 * nb:       # the same steps streamed (distinct batches back to back, stragglers handed over)
            ref_sync = sync_reference(ctx, eng, W, nb, a.stream_batches)
            JS = timed_stream_job(ctx, eng, W, nb, nb, J["steps"], 2, a.stream_batches, a.stream_contexts, ref_sync)
        r_stream = sweep_roofline(JS) if JS is not None else None
        best = JS if JS is not None else J               # (the streamed job whenever it was run: one methodology for every configuration and round)
        r = r_stream if best is JS else r_sync
        eng.close()
        parity = None
        if a.cpu_seconds > 0:
            parity, _ = parity_leg(b, W, ctx.G, res_np, nb, seconds=min(4.0, a.cpu_seconds / 3.0), threads=cpu_quota() or os.cpu_count() or 8)
        tol = None
        if a.arith == 0 and not a.no_tolerance_mode:
            tol = tolerance_block(ctx, a, prm, W, nb, nb, J["steps"], 2, res_np, JS is not None, min(2.0, a.cpu_seconds / 6.0) if a.cpu_seconds > 0 else 0.0, min_seconds=a.other_seconds, b=b)
        N = az * 64
        out[name] = {"workload": f"{baseline_config_name(b, N)}: {nb} synthetic HDL-64E scan pairs ({N} pts per cloud), ndt_pca, {b.resolution} m voxels, {b.mode.upper()}, "
                                 "eps 0.01, max_iter 64; one step = voxelise every target + align every pair",
                     "value": round(nb * J["steps"] / best["dt"], 2), "unit": "registrations/s", "steps": J["steps"], "warmup": 2,
                     "ms_per_step": round(1e3 * best["dt"] / J["steps"], 3), "timed_s": round(best["dt"], 3),
                     "value_mode": "streamed" if best is JS else "synchronous",
                     "value_tolerance_mode": None if tol is None else tol["value_tolerance_mode"], "tolerance_mode": tol,
                     "value_synchronous": round(nb * J["steps"] / J["dt"], 2), "ms_per_step_synchronous": round(1e3 * J["dt"] / J["steps"], 3),
                     "value_streamed": None if JS is None else round(nb * J["steps"] / JS["dt"], 2), "roofline_frac_streamed": None if JS is None else r_stream["frac"],
                     "roofline_frac_synchronous": r_sync["frac"], "avg_launch_us_synchronous": r_sync["avg_launch_us"],
                     "stream": None if JS is None else {k: JS[k] for k in ("n_batches", "n_contexts", "bit_identical_to_synchronous", "launches", "pairs_handed_over", "batches_rerun", "launches_that_gave_up")},
                     "mean_iterations": round(float(res_np["it"].mean()), 2), "converged": int(res_np["conv"].sum()),
                     "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us", "alg_bytes_per_launch",
                                                   "hits_per_point", "flops", "build_ms_per_step", "update_ms_per_step", "sweep_ms_per_step", "build_frac",
                                                   "launch_slots", "reserved_slots_for_the_next_build", "frac_of_its_slots") if k in r},
                     "roofline_valu": valu_roofline(f"{nb}x{N}:{b.variant}:{b.mode}:{b.resolution}"),
                     "parity": parity}
        attach_traffic(out[name]["roofline"], f"{nb}x{N}:{b.variant}:{b.mode}:{b.resolution}", r_sync["avg_launch_us"], None)
    del W5
    return {"configs": out, "seconds": round(time.perf_counter() - t0, 2), "input_generation_s": round(gen5, 2),
            "what": "same timed-step definition as the headline (barrier-free single rank, HIP events inside the engine for the sweep's roofline); "
                    "never `value`: the headline stays BASELINE config 3"}


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a, sys.argv[1:])                  # (does not return: the process becomes torch.distributed.run)
        raise SystemExit("bench.py: the launcher returned")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one process per GPU; LV_SLAM_BENCH_BACKEND=gloo lets several ranks share one GPU for a functional check of
    # the N>1 path on a single-GPU box (the driver's scaling runs use the default: nccl = RCCL over xGMI)
    backend = os.environ.get("LV_SLAM_BENCH_BACKEND", "nccl")
    visible = torch.cuda.device_count()
    if backend == "nccl" and visible < world:
        raise SystemExit(f"{world} ranks but only {visible} GPU(s) visible: one rank per GPU, never two ranks on one device under the RCCL backend")
    local = local % visible if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pinned_to = pin_to_gpu_numa_node(local)
    dist = None
    # LV_SLAM_BENCH_FORCE_DIST=1: a single rank still goes through the process group, the device-packed records and the RCCL
    # all-gather -- the N > 1 code path on a one-GPU box (tests/test_bench_gpu.py)
    use_dist = world > 1 or bool(os.environ.get("LV_SLAM_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")   # (torch.distributed.run sets both; the forced single-rank mode may not)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from lv_slam_amd import ndt, synth
    from lv_slam_amd import dist as shard

    if a.prefiltered:
        if world > 1:
            raise SystemExit("--prefiltered is a single-GPU row")
        prefiltered_row(a, ndt, synth, dev, local)
        return

    ctx = Ctx()
    ctx.rank, ctx.world, ctx.local, ctx.dev, ctx.dist, ctx.backend, ctx.use_dist = rank, world, local, dev, dist, backend, use_dist
    ctx.on_dev, ctx.shard, ctx.ndt, ctx.G = backend == "nccl", shard, ndt, synth.default_guess()
    ctx.stream_reserve = a.stream_reserve
    G = ctx.G

    strong = a.total_pairs > 0
    total = a.total_pairs if strong else a.pairs * world
    if total < world:
        raise SystemExit("fewer pairs than ranks")
    # The config-4 block (BASELINE config 4 as worded: ONE fixed job of 4,541 pairs sharded round-robin over the N ranks, strong
    # scaling) rides along with the default weak-scaling line, so that a driver that only ever passes --gpus N still measures it.
    # Both jobs give pair i to rank i mod N, so a rank's pairs of either job are a prefix of the same sequence rank, rank + N, ...:
    # the clouds are generated once, for the longer of the two prefixes.
    c4_total = 0 if (strong or a.config4_pairs <= 0 or a.config4_pairs < world) else a.config4_pairs
    pair_ids = shard.shard_pairs(total, rank, world)           # round-robin shard of the global pair index space
    c4_ids = shard.shard_pairs(c4_total, rank, world) if c4_total else []
    # the streamed job rotates through --stream-batches DISTINCT batches of the same size: more of the same sequence
    want_stream = not a.no_stream and not a.prefiltered
    st_ids = shard.shard_pairs(total * max(1, a.stream_batches), rank, world)[:len(pair_ids) * max(1, a.stream_batches)] if (want_stream and not a.kitti_dir) else pair_ids
    all_ids = max((pair_ids, c4_ids, st_ids), key=len)
    assert all_ids[:len(pair_ids)] == pair_ids and all_ids[:len(c4_ids)] == c4_ids and all_ids[:len(st_ids)] == st_ids
    B = len(pair_ids)
    # ---- inputs, resident in HBM before any timed region: [pair][3][pitch] SoA
    if a.kitti_dir:
        W, t_gen = load_kitti(ctx, a, ndt, all_ids)
    else:
        W, t_gen = generate_synthetic(ctx, synth, all_ids, a.azimuth)
    N = W["max_points"]
    # every rank's generation time, in the line (eight ranks of a node share the host's CPUs: the scaling run's one CPU-side cost)
    t_gen_ranks = [round(t_gen, 2)]
    if dist is not None:
        tg = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tg[rank] = t_gen
        dist.all_reduce(tg, op=dist.ReduceOp.SUM)
        t_gen_ranks = [round(float(v), 2) for v in tg.tolist()]

    prm = ndt.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=MODES[a.mode],
                             variant=1 if a.variant == "pca" else 0)
    eng = ndt.Engine(prm, device=local)
    eng.set_option(ndt.OPT_F32_SUM_ORDER, a.f32_sum_order)
    eng.set_option(ndt.OPT_ARITH, a.arith)

    # ---- the headline job
    J = timed_job(ctx, eng, W, B, total, a.steps, a.warmup)
    res, res_np, guesses, dt, steps = J["res"], J["res_np"], J["guesses"], J["dt"], J["steps"]
    # SURVEY 8(d) asks for the rate with and without setInputTarget: the same pairs again against the now-resident grids
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        eng.batch_align_raw(guesses, res)
    torch.cuda.synchronize()
    dt_resident = time.perf_counter() - t1
    # leaf statistics of the workload's targets (what the derivative sweep's cost depends on besides the point count)
    lv = [eng.get_voxels(k) for k in range(min(B, 8))]
    leaf_stats = {"targets_sampled": len(lv), "searchable_leaves_per_target_mean": round(float(np.mean([len(v) for v in lv])), 1),
                  "points_per_leaf_mean": round(float(np.mean([v["n"][v["n"] > 0].mean() for v in lv])), 2),
                  "largest_leaf_points": int(max(v["n"].max() for v in lv))}
    head_res = np.frombuffer(res, dtype=np.uint8).copy()       # (the config-4 block below re-binds the engine)
    res_np = np.frombuffer(head_res, dtype=RES_DT)

    # ---- the same job streamed: distinct batches submitted back to back, a launch's stragglers finishing under the next batch
    JS = None
    if want_stream and len(W["ids"]) >= 2 * B:
        ref_sync = sync_reference(ctx, eng, W, B, a.stream_batches)
        assert bytes(ref_sync[0]) == bytes(head_res), "the synchronous align is not deterministic"
        JS = timed_stream_job(ctx, eng, W, B, total, steps, a.warmup, a.stream_batches, a.stream_contexts, ref_sync)

    # ---- the drop-in path: host AoS clouds (pcl::PointXYZI records) in, PCIe inclusive -- rank 0's pairs
    host_path = None
    if rank == 0 and not a.no_host_clouds and (a.host_clouds or world == 1) and W["data"] == "synthetic":
        host_path, hres = host_clouds_leg(a, ndt, prm, local, W["T"][:B], W["S"][:B], guesses, B, N, steps=max(2, min(steps, 40) // 2), uploaders=a.uploaders or None)
        # same bits as the device-resident run of the same pairs
        nh = len(np.frombuffer(hres[0], dtype=RES_DT))      # the leg's pairs: the first min(B, 271) of the batch
        host_path["pairs_per_batch"] = nh
        host_path["bit_identical_to_device_resident_run"] = bool(all(
            np.array_equal(np.frombuffer(hres[i], dtype=RES_DT)["final"], res_np["final"][:nh]) and np.array_equal(np.frombuffer(hres[i], dtype=RES_DT)["score"], res_np["score"][:nh])
            for i in range(2)))
        # ... and streamed through ONE engine (mi355ndt_stream_submit_host): staging and PCIe of the next batch under the launch of the current one
        hs, hsres = host_clouds_stream_leg(a, ndt, prm, local, W["T"][:B], W["S"][:B], guesses, B, N, steps=max(4, min(steps, 40) // 2), threads=a.uploaders or None)
        hs["bit_identical_to_device_resident_run"] = bool(bytes(hsres) == bytes(head_res[:len(hsres)]))
        host_path["two_synchronous_engines_registrations_per_s"] = host_path["registrations_per_s"]
        host_path["streamed"] = hs

    # ---- latency mode: the nodelet's own per-frame loop on a drive (never `value`)
    seq_leg = None
    if rank == 0 and world == 1 and a.seq_frames > 1 and W["data"] == "synthetic":
        seq_leg = sequential_leg(a, ndt, local, dev, a.seq_frames)

    # ---- BASELINE config 4 beside it: the fixed 4,541-pair job, strong-scaled over the same ranks
    cfg4 = None
    if c4_total:
        J4 = timed_job(ctx, eng, W, len(c4_ids), c4_total, max(3, steps // 4) if a.steps is not None else None, min(a.warmup, 2))
        r4 = sweep_roofline(J4)
        cfg4 = {"workload": f"BASELINE config 4: ONE job of {c4_total} {'KITTI' if W['data'] == 'kitti' else 'synthetic HDL-64E'} scan pairs ({N} pts per cloud) sharded round-robin over {world} GPU(s), "
                            f"ndt_{a.variant}, {a.resolution} m voxels, {a.mode.upper()}; one step = voxelise + align this rank's shard + the pose all-gather of all {c4_total} records",
                "scaling": "strong", "pairs_total": c4_total, "pairs_rank0": len(c4_ids), "n_gpus": world, "steps": J4["steps"],
                "value": round(c4_total * J4["steps"] / J4["dt"], 2), "unit": "registrations/s", "ms_per_step": round(1e3 * J4["dt"] / J4["steps"], 3),
                "gather_ms_per_step": J4["gather_ms_per_step"], "gather_check": J4["gather_check"],
                "roofline_frac": r4["frac"], "avg_sweep_launch_us": r4["avg_launch_us"], "build_ms_per_step": r4["build_ms_per_step"],
                "update_ms_per_step": r4["update_ms_per_step"], "sweep_ms_per_step": r4["sweep_ms_per_step"],
                "mean_iterations": round(float(J4["res_np"]["it"].mean()), 2), "converged": int(J4["res_np"]["conv"].sum()), "parity": None}
        if rank == 0 and a.cpu_seconds > 0 and world == 1:
            cfg4["parity"], _ = parity_leg(a, W, G, J4["res_np"], len(c4_ids), seconds=a.cpu_seconds / 2.0, threads=cpu_quota() or os.cpu_count() or 8)

    # ---- the same jobs in the tolerance arithmetic (never `value`): every rank runs them (they contain the same barriers / gathers)
    tol_head, tol_c4 = None, None
    tol_ok = a.arith == 0 and not a.no_tolerance_mode and a.mode in ("direct1", "direct7")
    if tol_ok:
        osec = (a.cpu_seconds / 4.0) if (rank == 0 and world == 1) else 0.0
        tol_head = tolerance_block(ctx, a, prm, W, B, total, steps, min(a.warmup, 3), res_np, JS is not None, osec)
        if c4_total:
            tol_c4 = tolerance_block(ctx, a, prm, W, len(c4_ids), c4_total, J4["steps"], 2, np.frombuffer(np.frombuffer(J4["res"], dtype=np.uint8).copy(), dtype=RES_DT), False, osec)

    pg = {"world_size": dist.get_world_size() if dist is not None else 1, "backend": dist.get_backend() if dist is not None else None,
          "launcher": os.environ.get("LV_SLAM_BENCH_LAUNCHER") or ("torch.distributed.run (external)" if "TORCHELASTIC_RUN_ID" in os.environ else "plain process"),
          "devices_visible": visible, "device_of_rank0": torch.cuda.get_device_name(local),
          "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if (dist is not None and backend == "nccl") else None,
          "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if rank != 0:
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    value_sync = total * steps / dt
    value_stream = total * JS["steps"] / JS["dt"] if JS is not None else None
    # ONE methodology, every round and configuration: `value` is the STREAMED job whenever it was run (BASELINE config 3: "streamed through 1 GPU"; distinct
    # batches back to back, every step's results on the host inside the timed region), else the synchronous one; `value_mode` says which, and both
    # rates are always in the line (value_synchronous, value_streamed).  (Rounds 4-5 printed the faster of the two.)
    streamed_wins = value_stream is not None
    value = value_stream if streamed_wins else value_sync
    its = res_np["it"].astype(np.float64)
    sweeps = res_np["sweeps"].astype(np.float64)
    # ---- roofline of the dominant kernel (derivative sweep): algorithmic bytes / HIP-event time
    wkey = f"{a.pairs if not strong else B}x{N}:{a.variant}:{a.mode}:{a.resolution}"
    roof_sync = sweep_roofline(J)
    roof_stream = sweep_roofline(JS) if JS is not None else None
    roof = roof_stream if streamed_wins else roof_sync
    attach_traffic(roof, wkey, roof_sync["avg_launch_us"], a.traffic)
    roof_valu = valu_roofline(wkey)
    # what the figure is, in the line itself: `achieved` prices the ALGORITHMIC bytes of SURVEY 8(d) (every voxel record a point evaluates
    # counted as read) against the HBM peak; the voxel records are re-used out of L2 / MALL, so the PHYSICAL HBM rate is `traffic` /
    # launch time, and the unit the sweep is limited by is the vector ALU (`roofline_valu`)
    roof["what"] = ("achieved = algorithmic bytes (SURVEY 8(d): 12 + 4*neighbours + 64*hits per point) / launch time: a cache-resident working set priced "
                    "against HBM, not HBM traffic; physical_hbm_frac is the measured HBM share; the binding unit is VALU issue (roofline_valu.active_frac)")
    if roof.get("reserved_slots_for_the_next_build"):
        roof["what"] += (f"; THIS launch runs on {roof['launch_slots']} of {roof['launch_slots'] + roof['reserved_slots_for_the_next_build']} workgroup slots with the next batch's "
                         "target build beside it on the others (stream mode, MI355NDT_OPT_STREAM_RESERVE): it is slower by the slots it gives away and the step is shorter "
                         "by the whole build -- frac_of_its_slots is the same rate per slot it had, roofline_synchronous.frac the same kernel with the GPU to itself")

    cpu, parity = (None, None)
    if a.cpu_seconds > 0 and world == 1:          # the CPU leg runs on rank 0 of the single-GPU run only
        cpu, parity = cpu_leg(a, W, G, res_np, B)

    # ---- the other BASELINE configurations, in the same line (single-GPU run of the default workload family)
    others = None
    if world == 1 and not a.no_other_configs and not strong and W["data"] == "synthetic" and a.variant == "omp" and a.mode == "direct7" and a.resolution == 1.0:
        others = other_configs_block(ctx, a, synth, W)

    kind = "KITTI" if W["data"] == "kitti" else "synthetic HDL-64E"
    out = {
        "metric": "NDT registrations/sec (64k-pt Velodyne pairs)", "value": round(value, 2), "unit": "registrations/s",
        "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": round(1e3 * (JS["dt"] if streamed_wins else dt) / steps, 3),
        "value_mode": "streamed" if streamed_wins else "synchronous",
        "value_synchronous": round(value_sync, 2), "ms_per_step_synchronous": round(1e3 * dt / steps, 3),
        "value_streamed": None if value_stream is None else round(value_stream, 2), "ms_per_step_streamed": None if JS is None else round(1e3 * JS["dt"] / steps, 3),
        "value_tolerance_mode": None if tol_head is None else tol_head["value_tolerance_mode"], "tolerance_mode": tol_head,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32 terms, f64 accumulation" if a.arith == 0 else "f32 terms and f32 sums per 512-point work item (tolerance arithmetic, --arith 1), f64 accumulation from there on",
        "data": W["data"],
        "config": {"workload": (f"BASELINE config 4: {total} {kind} scan pairs sharded round-robin over {world} GPU(s) " if strong else
                                f"{baseline_config_name(a, N) if W['data'] == 'synthetic' else 'KITTI seq ' + os.path.basename(os.path.dirname(os.path.abspath(a.kitti_dir).rstrip('/')))}: "
                                f"batch of {a.pairs} {kind} scan pairs per GPU ") +
                               f"({N} pts per cloud{' at most' if W['data'] == 'kitti' else ''}), ndt_{a.variant}, {a.resolution} m voxels, {a.mode.upper()}, eps 0.01, max_iter 64; "
                               "one step = voxelise every target + align every pair (+ RCCL pose all-gather when N>1)",
                   "mode": ("synchronous: batch_build_targets + batch_align, one batch at a time" if JS is None else
                            f"streamed: {JS['n_batches']} distinct batches of {B} pairs submitted back to back through mi355ndt_stream_* ({JS['n_contexts']} batches resident; a launch hands "
                            "its last unfinished pairs to the next launch; every step's results on the host inside the timed region; bit-identical to the synchronous align); "
                            "value_synchronous = the same steps through batch_build_targets + batch_align, one batch at a time"),
                   "stream": None if JS is None else {k: JS[k] for k in ("n_batches", "n_contexts", "bit_identical_to_synchronous", "launches", "pairs_handed_over", "batches_rerun", "launches_that_gave_up")},
                   "pairs_total": total, "pairs_rank0": B, "points_per_cloud": N, "neighbor_mode": a.mode, "variant": a.variant,
                   "resolution_m": a.resolution, "f32_sum_order": a.f32_sum_order, "arith": a.arith, "sharding": "pair i -> rank i mod N (round-robin)",
                   "mean_iterations": round(float(its.mean()), 2), "max_iterations_seen": int(its.max()),
                   "mean_sweeps_per_align": round(float(sweeps.mean()), 2),
                   "converged": int(res_np["conv"].sum()),
                   "rank0_registrations_per_s_resident_targets": round(B * steps / dt_resident, 1),
                   "target_leaf_statistics": leaf_stats,
                   "steps_chosen_by": "--steps" if a.steps is not None else "timed region sized to >= 0.5 s",
                   "input_generation_s": round(max(t_gen_ranks), 2), "input_generation_s_per_rank": t_gen_ranks, "pairs_generated_per_rank": len(all_ids), "inputs": W["generated_on"],
                   "mean_points_per_source": round(float(np.mean(W["scnt"][:B])), 1)},
        "world_size": pg["world_size"], "process_group": pg,
        "gather_ms_per_step": (JS if streamed_wins else J)["gather_ms_per_step"], "gather_ms_per_step_streamed": None if JS is None else JS["gather_ms_per_step"],
        "roofline_synchronous": None if JS is None else {k: roof_sync[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "alg_bytes_per_launch", "build_ms_per_step", "sweep_ms_per_step")},
        "roofline_streamed": None if JS is None else {k: roof_stream[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "alg_bytes_per_launch", "build_ms_per_step", "sweep_ms_per_step")},
        "roofline": roof, "roofline_valu": roof_valu, "cpu_baseline": cpu, "parity": parity, "gather_check": (JS if streamed_wins and JS["gather_check"] is not None else J)["gather_check"], "gather_check_streamed": None if JS is None else JS["gather_check"],
        "config4": cfg4, "other_configs": others,
    }
    if cfg4 is not None and tol_c4 is not None:
        cfg4["value_tolerance_mode"] = tol_c4["value_tolerance_mode"]
        cfg4["tolerance_mode"] = tol_c4
    if seq_leg is not None:
        out["value_sequential"] = seq_leg["frames_per_s"]
        out["sequential"] = seq_leg
    if host_path is not None:
        host_path["process_pinned_to"] = pinned_to
        out["value_host_clouds"] = max(host_path["registrations_per_s"], host_path["streamed"]["registrations_per_s"])
        out["value_host_clouds_mode"] = "streamed (mi355ndt_stream_submit_host, one engine)" if host_path["streamed"]["registrations_per_s"] >= host_path["registrations_per_s"] else "two synchronous engines"
        out["host_clouds"] = host_path
    print(json.dumps(out), flush=True)
    eng.close()                                   # release HIP objects before interpreter teardown (profilers hook exit)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
