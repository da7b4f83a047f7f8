#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for c in r04 6b5adab 8bb3c71 d8df312 487d3df 607f0a1; do
  (cd $R/tools/ab_$c && timeout 600 python bench.py --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_synchronous') or d['roofline']
print('$c', d.get('value_synchronous', d['value']), r['avg_launch_us'])")
done; done
