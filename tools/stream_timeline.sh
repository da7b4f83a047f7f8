#!/bin/bash
# Kernel + copy timeline of the last steps of bench.py's STREAMED job (GPU box, through gpurun): every kernel and copy of the engine in start
# order with its duration and the idle time of the stream in front of it -- how the streamed step was taken apart (DESIGN.md 4.2b,
# docs/experiments.md 10d).   usage: tools/stream_timeline.sh <tag> [bench.py arguments...]   -> gpurun_out/<tag>/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --no-other-configs --steps 12 --warmup 2 "$@" > $O/bench.json 2> $O/kt.log
python - <<PY > $O/timeline.txt
import csv, glob, re
f = glob.glob("$O/kt/**/*_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
mc = glob.glob("$O/kt/**/*_memory_copy_trace.csv", recursive=True)
cps = sorted(csv.DictReader(open(mc[0])), key=lambda r: int(r["Start_Timestamp"])) if mc else []
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.split(r"[<(]", r["Kernel_Name"].replace("void ", ""))[0]) for r in rows
      if re.search(r"k_|rocclr", r["Kernel_Name"]) and "at::" not in r["Kernel_Name"]]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:24]) for r in cps]
ev.sort()
first = [i for i, e in enumerate(ev) if e[2] == "k_stream_inputs"]        # a streamed batch starts with its input kernel
per = [(ev[first[i + 1]][0] - ev[first[i]][0]) / 1e3 for i in range(len(first) - 1)]
print("streamed job: periods between consecutive batches (us):", [round(p) for p in per])
start = first[-3]
t0 = ev[start][0]; prev = t0
print(f"{'kernel / copy':34s} {'start_us':>10s} {'dur_us':>9s} {'idle_before_us':>14s}")
for s, e, n in ev[start:]:
    print(f"{n[:34]:34s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} {(s - prev) / 1e3:14.2f}")
    prev = max(prev, e)
PY
rm -rf $O/kt
cat $O/timeline.txt
