#!/bin/bash
# SQ / TCP counters of one engine kernel over a short bench.py run (GPU box, through gpurun).  Counters are collected in their
# own passes, without any trace domain.   usage: tools/pmc_kernel.sh <kernel regex> <tag> [bench.py arguments...]
R=${GRAFT_REPO_ROOT:-/root/repo}
KRE=$1; TAG=$2; shift; shift
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$KRE" --output-format csv -d $O/p$k -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --no-other-configs --steps 2 --warmup 1 "$@" > $O/p$k.log 2>&1
  k=$((k+1))
done
python - <<PY | tee $O/summary.txt
import csv, glob, collections, re
agg = collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True)):
    byname = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = re.sub(r"^void ", "", r.get("Kernel_Name", ""))
        kn = re.split(r"[<(]", kn)[0]
        byname[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for n, v in byname.items():
        agg[n] = (sum(v) / len(v), len(v))
print("kernel regex: $KRE   (mean per dispatch)")
kernels = sorted({k for k, _ in agg})
for kn in kernels:
    if len(kernels) > 1:
        print(f"--- {kn}")
    for (k, n), (v, c) in agg.items():
        if k == kn:
            print(f"{n:32s} {v:18.0f}   ({c} dispatches)")
PY
rm -rf $O/p*/
