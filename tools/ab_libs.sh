#!/bin/bash
# times the synchronous config-3 job with several builds of the library (MI355NDT_LIB) on one box: usage tools/ab_libs.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  (cd tools/ab_r04 && timeout 600 python bench.py --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('r04', d['value'], r['avg_launch_us'])")
  for L in "$@"; do
    MI355NDT_LIB=$R/lv_slam_amd/$L timeout 600 python bench.py --no-stream ${BENCH_ARGS} --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$L', d['value'], r['avg_launch_us'])"
  done
done
