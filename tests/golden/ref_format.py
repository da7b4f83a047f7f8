"""The file format of the parity pin kit (tools/pin_reference/dump_golden.cpp writes it, tests/test_reference_golden.py reads it).

tests/golden/pin/cases.txt, one case per line:
    name variant mode resolution step_size outlier_ratio trans_epsilon max_iterations target.bin source.bin g0 .. g15
(variant 0 = pclomp, 1 = pclpca; mode = NeighborSearchMethod enum value, ndt_omp.h:51-56; clouds = raw little-endian f32 x,y,z triples;
guess = 4x4 f32 column-major).

tests/golden/ref_<name>.bin ("NDTREF01", little endian):
    char[8] magic | int32 variant, mode, n_target, n_source, max_iterations, n_leaves | uint32 flags | float32 resolution
    | float64 step_size, outlier_ratio, trans_epsilon
    | n_leaves x { int64 idx; int32 nr_points; int32 weight; float64 mean[3], cov[9], icov[9], evals[3] }     (std::map order; matrices row-major)
    | float64 p[6], score, g[6], H[36]                                                                         (one computeDerivatives sweep at the guess)
    | float32 final[16], last_increment[16] (column-major) | int32 iterations, converged | float64 trans_probability, calculate_score
flags: 1 = the sweep block is filled, 2 = the leaf list holds EVERY occupied cell (else only the searchable ones), 4 = cov / evals are filled.
"""
from __future__ import annotations

import os
import numpy as np

MAGIC = b"NDTREF01"
LEAF_DT = np.dtype([("idx", "<i8"), ("n", "<i4"), ("weight", "<i4"), ("mean", "<f8", 3), ("cov", "<f8", 9), ("icov", "<f8", 9), ("evals", "<f8", 3)])
HEAD_DT = np.dtype([("magic", "S8"), ("variant", "<i4"), ("mode", "<i4"), ("n_target", "<i4"), ("n_source", "<i4"), ("max_iterations", "<i4"),
                    ("n_leaves", "<i4"), ("flags", "<u4"), ("resolution", "<f4"), ("step_size", "<f8"), ("outlier_ratio", "<f8"), ("trans_epsilon", "<f8")])
TAIL_DT = np.dtype([("p", "<f8", 6), ("score", "<f8"), ("g", "<f8", 6), ("H", "<f8", 36), ("final", "<f4", 16), ("last_inc", "<f4", 16),
                    ("iterations", "<i4"), ("converged", "<i4"), ("trans_probability", "<f8"), ("calc_score", "<f8")])
assert LEAF_DT.itemsize == 208 and HEAD_DT.itemsize == 64 and TAIL_DT.itemsize == 544
HAS_SWEEP, ALL_LEAVES, HAS_COV = 1, 2, 4


def read_cases(pin_dir: str) -> list[dict]:
    out = []
    for line in open(os.path.join(pin_dir, "cases.txt")):
        w = line.split()
        if not w or w[0].startswith("#"):
            continue
        if len(w) != 26:
            raise ValueError("malformed case line: " + line)
        out.append(dict(name=w[0], variant=int(w[1]), mode=int(w[2]), resolution=float(w[3]), step_size=float(w[4]), outlier_ratio=float(w[5]),
                        trans_epsilon=float(w[6]), max_iterations=int(w[7]), target=w[8], source=w[9],
                        guess=np.array(w[10:26], np.float32).reshape(4, 4, order="F")))
    return out


def load_cloud(path: str) -> np.ndarray:
    return np.fromfile(path, "<f4").reshape(-1, 3)


def read_ref(path: str) -> dict:
    raw = open(path, "rb").read()
    h = np.frombuffer(raw[:HEAD_DT.itemsize], HEAD_DT)[0]
    if h["magic"] != MAGIC:
        raise ValueError(f"{path}: not an NDTREF01 file")
    nl = int(h["n_leaves"])
    o = HEAD_DT.itemsize
    leaves = np.frombuffer(raw[o:o + nl * LEAF_DT.itemsize], LEAF_DT).copy()
    o += nl * LEAF_DT.itemsize
    if len(raw) != o + TAIL_DT.itemsize:
        raise ValueError(f"{path}: {len(raw)} bytes, expected {o + TAIL_DT.itemsize}")
    t = np.frombuffer(raw[o:], TAIL_DT)[0]
    d = {k: h[k].item() for k in HEAD_DT.names if k != "magic"}
    d.update(leaves=leaves, p=t["p"].copy(), score=float(t["score"]), g=t["g"].copy(), H=t["H"].reshape(6, 6).copy(),
             final=t["final"].reshape(4, 4, order="F").copy(), last_inc=t["last_inc"].reshape(4, 4, order="F").copy(),
             iterations=int(t["iterations"]), converged=int(t["converged"]), trans_probability=float(t["trans_probability"]), calc_score=float(t["calc_score"]))
    return d


def write_ref(path: str, d: dict) -> None:
    """Python twin of dump_golden.cpp's writer (the format self-test writes oracle results with it)."""
    h = np.zeros(1, HEAD_DT)
    h["magic"] = MAGIC
    for k in HEAD_DT.names:
        if k not in ("magic", "n_leaves"):
            h[k] = d[k]
    leaves = np.ascontiguousarray(d["leaves"], LEAF_DT)
    h["n_leaves"] = len(leaves)
    t = np.zeros(1, TAIL_DT)
    t["p"], t["score"], t["g"], t["H"] = d["p"], d["score"], d["g"], np.asarray(d["H"], np.float64).reshape(36)
    t["final"] = np.asarray(d["final"], np.float32).ravel(order="F")
    t["last_inc"] = np.asarray(d["last_inc"], np.float32).ravel(order="F")
    t["iterations"], t["converged"], t["trans_probability"], t["calc_score"] = d["iterations"], d["converged"], d["trans_probability"], d["calc_score"]
    with open(path, "wb") as f:
        f.write(h.tobytes() + leaves.tobytes() + t.tobytes())
