#!/bin/bash
# A/B of the one-launch align's counters: round-4 tree (tools/ab_r04) against HEAD, same box, same workload (sync path, config 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --no-other-configs --steps 3 --warmup 1"
k=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  for tree in r04 head; do
    if [ $tree = r04 ]; then D=$R/tools/ab_r04; X=""; else D=$R; X="--no-stream"; fi
    (cd $D && timeout 300 rocprofv3 --pmc $set --kernel-include-regex 'k_align_async' --output-format csv -d $O/$tree$k -- python $D/bench.py $ARGS $X > $O/$tree$k.log 2>&1)
  done
  k=$((k+1))
done
python - <<PY
import csv, glob, collections
for tree in ("r04", "head"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob("$O/%s*/**/*_counter_collection.csv" % tree, recursive=True)):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for n, v in by.items():
            agg[n] = (sum(v) / len(v), len(v))
    print(tree)
    for n, (v, c) in agg.items():
        print(f"  {n:28s} {v:18.0f} ({c})")
PY
rm -rf $O/*/
