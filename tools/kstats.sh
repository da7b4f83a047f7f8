#!/bin/bash
# Per-kernel times of one bench.py run under rocprofv3 (runs on the GPU box through gpurun).
# usage: tools/kstats.sh <tag> [bench.py arguments...]   -> gpurun_out/<tag>/kstats.txt (engine kernels only) + the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 --config4-pairs 0 --seq-frames 0 --no-other-configs --no-tolerance-mode "$@" > $O/bench.json 2> $O/kt.log
python - <<PY > $O/kstats.txt
import csv, glob
f = glob.glob("$O/kt/**/*_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s}")
for r in rows:
    n = r["Name"]
    if n.startswith("void "): n = n[5:]
    if any(k in n for k in ("k_", "rocprim", "rocclr", "hipcub")) and "at::" not in n:
        print(f"{n[:60]:60s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:10.1f} {float(r['AverageNs'])/1e3:9.2f}")
PY
rm -rf $O/kt/*/*.db
cat $O/kstats.txt; tail -c 1500 $O/bench.json
