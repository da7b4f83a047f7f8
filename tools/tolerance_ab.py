#!/usr/bin/env python3
"""GPU box: the exact arithmetic and the tolerance arithmetic (MI355NDT_OPT_ARITH = 1) side by side on one box, one workload:
step time (build + one-launch align, HIP events inside the engine), and the tolerance mode's results against the exact ones pair by pair
(iteration flips, pairs beyond 1e-4 m / 1e-5 rad, largest deltas).  usage: tools/tolerance_ab.py [--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128]"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from lv_slam_amd import ndt, synth
from lv_slam_amd import dist as shard

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="omp"); ap.add_argument("--mode", default="direct7"); ap.add_argument("--resolution", type=float, default=1.0)
ap.add_argument("--azimuth", type=int, default=1024); ap.add_argument("--pairs", type=int, default=271); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--first", type=int, default=0, help="first pair index")
ap.add_argument("--only", default="", help="'fast' / 'exact': time one arithmetic only")
a = ap.parse_args()
ctx = bench.Ctx()
ctx.rank, ctx.world, ctx.local, ctx.dev, ctx.dist, ctx.use_dist = 0, 1, 0, torch.device("cuda", 0), None, False
ctx.on_dev, ctx.shard, ctx.ndt, ctx.G = True, shard, ndt, synth.default_guess()
torch.cuda.set_device(0)
W, tgen = bench.generate_synthetic(ctx, synth, list(range(a.first, a.first + a.pairs)), a.azimuth)
prm = ndt.default_params(resolution=a.resolution, trans_epsilon=0.01, max_iterations=64, neighbor_mode=bench.MODES[a.mode], variant=1 if a.variant == "pca" else 0)
out = {"lib": os.path.basename(ndt.LIB_PATH), "workload": f"{a.pairs}x{a.azimuth * 64} {a.variant} {a.mode} {a.resolution}", "gen_s": round(tgen, 1)}
res, vox = {}, {}
for name, arith in (("exact", 0), ("fast", 1)):
    if a.only and a.only != name:
        continue
    eng = ndt.Engine(prm, device=0)
    eng.set_option(ndt.OPT_ARITH, arith)
    J = bench.timed_job(ctx, eng, W, a.pairs, a.pairs, a.steps, 3)
    r = bench.sweep_roofline(J)
    res[name] = np.frombuffer(np.frombuffer(J["res"], dtype=np.uint8).copy(), dtype=bench.RES_DT)
    vox[name] = eng.get_voxels(0)
    out[name] = {"reg_s": round(a.pairs * J["steps"] / J["dt"], 1), "ms_per_step": round(1e3 * J["dt"] / J["steps"], 3), "launch_us": r["avg_launch_us"], "build_ms": r["build_ms_per_step"],
                 "frac": r["frac"], "mean_it": round(float(res[name]["it"].mean()), 3), "hits_per_point": r["hits_per_point"]}
    eng.close()
if len(res) == 2:
    E, F = res["exact"], res["fast"]
    dts, drs = [], []
    for k in range(a.pairs):
        dt_, dr_ = bench.se3_err(E["final"][k].reshape(4, 4).T, F["final"][k].reshape(4, 4).T)
        dts.append(dt_); drs.append(dr_)
    dts, drs = np.array(dts), np.array(drs)
    flips = int((E["it"] != F["it"]).sum())
    out["fast_vs_exact"] = {"pairs": a.pairs, "iteration_flips": flips, "beyond_tolerance": int(((dts >= 1e-4) | (drs >= 1e-5)).sum()), "max_dtrans_m": float(dts.max()), "max_drot_rad": float(drs.max()),
                            "median_dtrans_m": float(np.median(dts)), "p99_dtrans_m": float(np.percentile(dts, 99)), "converged_equal": int((E["conv"] == F["conv"]).sum()),
                            "max_rel_dscore": float(np.max(np.abs(E["score"] - F["score"]) / np.maximum(1e-300, np.abs(E["score"])))),
                            "speedup": round(out["fast"]["reg_s"] / out["exact"]["reg_s"], 3)}
if len(vox) == 2:
    A, B = vox["exact"], vox["fast"]
    same_shape = len(A) == len(B) and bool(np.array_equal(A["idx"], B["idx"])) and bool(np.array_equal(A["n"], B["n"]))
    out["voxels_pair0"] = {"leaves": len(A), "same_cells_and_counts": same_shape}
    if same_shape:
        for f in A.dtype.names:
            if A[f].dtype.kind == "f":
                d = np.abs(A[f].astype(np.float64) - B[f].astype(np.float64)); sc = np.maximum(1e-300, np.abs(A[f].astype(np.float64)))
                out["voxels_pair0"]["max_rel_" + f] = float(np.max(d / sc)) if d.size else 0.0
print(json.dumps(out), flush=True)
