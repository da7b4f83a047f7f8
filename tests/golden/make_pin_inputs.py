#!/usr/bin/env python3
"""Writes tests/golden/pin/: the input side of the parity pin kit (tools/pin_reference/README.md) -- the clouds of the committed
fixtures tests/golden/*.npz as raw f32 x,y,z triples, one full-size HDL-64E pair (65,536 points, synthetic pair 0, the pair of
BASELINE configs 1-2), and cases.txt naming every (class, neighbour mode, resolution, guess) the real pclomp:: / pclpca:: classes are
to be run on.  Run once; the outputs are committed.     python tests/golden/make_pin_inputs.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
PIN = os.path.join(HERE, "pin")
FIXTURES = ["omp_direct7_r1", "omp_direct1_r1", "omp_direct26_r2", "pca_direct7_r1", "pca_direct1_r05", "omp_kdtree_r1", "pca_kdtree_r1"]


def line(name, variant, mode, res, step, outlier, eps, maxit, tfile, sfile, guess):
    g = " ".join(repr(float(v)) for v in np.asarray(guess, np.float32).ravel(order="F"))
    return f"{name} {variant} {mode} {res!r} {step!r} {outlier!r} {eps!r} {maxit} {tfile} {sfile} {g}\n"


def main():
    os.makedirs(PIN, exist_ok=True)
    rows = ["# name variant mode resolution step_size outlier_ratio trans_epsilon max_iterations target.bin source.bin guess[16] (column-major)\n",
            "# variant: 0 = pclomp::NormalDistributionsTransform, 1 = pclpca::;  mode: KDTREE 0, DIRECT26 1, DIRECT7 2, DIRECT1 3 (ndt_omp.h:51-56)\n"]
    for name in FIXTURES:
        z = np.load(os.path.join(HERE, name + ".npz"))
        pv = z["params"]
        np.ascontiguousarray(z["target"], "<f4").tofile(os.path.join(PIN, name + "_target.bin"))
        np.ascontiguousarray(z["src_align"], "<f4").tofile(os.path.join(PIN, name + "_source.bin"))
        rows.append(line(name, int(pv[6]), int(pv[5]), float(pv[0]), float(pv[1]), float(pv[2]), float(pv[3]), int(pv[4]),
                         name + "_target.bin", name + "_source.bin", z["guess"]))
    from lv_slam_amd import synth
    tgt, src, _ = synth.make_pair(0, 1024)
    np.ascontiguousarray(tgt.numpy(), "<f4").tofile(os.path.join(PIN, "full_pair0_target.bin"))
    np.ascontiguousarray(src.numpy(), "<f4").tofile(os.path.join(PIN, "full_pair0_source.bin"))
    G = synth.default_guess()
    # BASELINE configs 1-2 (ndt_omp, 1.0 m, DIRECT7) and the live nodelet's registration (ndt_pca, 1.0 m, DIRECT1: scan_matching_odom_nodelet.cpp:109-119)
    rows.append(line("full_omp_direct7", 0, 2, 1.0, 0.1, 0.55, 0.01, 64, "full_pair0_target.bin", "full_pair0_source.bin", G))
    rows.append(line("full_pca_direct1", 1, 3, 1.0, 0.1, 0.55, 0.01, 64, "full_pair0_target.bin", "full_pair0_source.bin", G))
    open(os.path.join(PIN, "cases.txt"), "w").writelines(rows)
    print("wrote", len(rows) - 2, "cases,", sum(os.path.getsize(os.path.join(PIN, f)) for f in os.listdir(PIN)) // 1024, "KiB")


if __name__ == "__main__":
    main()
