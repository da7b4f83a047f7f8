#!/bin/bash
# SQ counters of full sweep launches (tools/sweep_only.py workload).  usage: tools/pmc_sweep.sh <mode> <variant>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$1_$2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT"; do
  MODE=$1 VARIANT=$2 timeout 300 rocprofv3 --pmc $set --kernel-include-regex k_sweep --output-format csv -d $O/p$k -- python $R/tools/sweep_only.py > $O/p$k.log 2>&1
  k=$((k+1))
done
python - <<PY
import csv, glob, collections
best = collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    byname = collections.defaultdict(list)
    for r in rows:
        byname[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, v in byname.items():
        best[n] = max(v)          # the full-batch launches are the largest
for n, v in best.items():
    print(f"{n:28s} {v:16.0f}")
PY
