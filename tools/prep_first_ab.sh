run() { timeout 600 python bench.py "$@" --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('tolerance_mode') or {}
print('prep_first=$MI355NDT_STREAM_PREP_FIRST $*', '| stream', d['value'], 'sync', d['value_synchronous'], 'ms', d['ms_per_step'], 'launch', r['avg_launch_us'], '| tol stream', t.get('value_tolerance_mode_streamed'))"; }
for rep in 1 2; do for pf in 0 1; do export MI355NDT_STREAM_PREP_FIRST=$pf
  run
  run --variant pca --mode direct1
  run --pairs 64
done; done
