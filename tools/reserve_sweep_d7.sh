#!/bin/bash
# config 3 (DIRECT7) over MI355NDT_OPT_STREAM_RESERVE, both arithmetics (GPU box): does the next batch's build hide under a VALU-bound launch?
mkdir -p gpurun_out/rs7
for r in ${RS:-0 64 128 192 256 384 0}; do
  timeout 300 python bench.py --stream-reserve $r --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('tolerance_mode') or {}
print('reserve $r', 'stream', d['value'], 'sync', d['value_synchronous'], 'ms', d['ms_per_step'], 'launch', r['avg_launch_us'], 'build', r['build_ms_per_step'], '| tol stream', t.get('value_tolerance_mode_streamed'), 'sync', t.get('value_tolerance_mode_synchronous'))"
done > gpurun_out/rs7/out.txt 2>&1
cat gpurun_out/rs7/out.txt
