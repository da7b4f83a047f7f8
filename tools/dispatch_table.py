"""Per-dispatch kernel durations of the LAST step of a rocprofv3 --kernel-trace run, in launch order.
usage: dispatch_table.py <trace dir> <kernel regex>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
pat = re.compile(sys.argv[2])
sel = [r for r in rows if pat.search(r["Kernel_Name"])]
# the last step = everything after the last k_minmax (the first kernel of a target build)
last = max((i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_minmax(")), default=0)
t0 = int(rows[last]["Start_Timestamp"])
print(f"{'kernel':40s} {'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s}")
prev_end = t0
for r in rows[last:]:
    if not pat.search(r["Kernel_Name"]):
        continue
    name = re.split(r"[<(]", r["Kernel_Name"].replace("void ", ""))[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{name[:40]:40s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {(s - prev_end) / 1e3:8.2f}")
    prev_end = e
