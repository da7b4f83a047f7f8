"""Median duration / gap of the kernels of the sequence pump from a rocprofv3 --kernel-trace directory."""
import csv, glob, re, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if re.search(r"k_seq_update|k_sweep", r["Kernel_Name"])]
d = {"k_seq_update": [], "k_sweep": []}
gaps = []
for a, b in zip(sel, sel[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for r in sel:
    k = "k_seq_update" if "k_seq_update" in r["Kernel_Name"] else "k_sweep"
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = np.array(v)
    print(f"{k:14s} n {len(v):6d}  median {np.median(v):7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")
g = np.array(gaps)
print(f"gap between consecutive pump kernels: median {np.median(g):.2f} us, p90 {np.percentile(g, 90):.2f}")
