run() { timeout 600 python bench.py "$@" --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('tolerance_mode') or {}
print('$*', '| stream', d['value'], 'sync', d['value_synchronous'], 'ms', d['ms_per_step'], 'launch', r['avg_launch_us'], '| tol stream', t.get('value_tolerance_mode_streamed'), 'sync', t.get('value_tolerance_mode_synchronous'))"; }
for p in 64 128; do
for r in 64 128 192; do run --pairs $p --stream-reserve $r; done
for r in 96 160 224; do run --pairs $p --variant pca --mode direct1 --stream-reserve $r; done
done
for r in 64 128; do run --pairs 16 --stream-reserve $r; done
for r in 96 160; do run --pairs 16 --variant pca --mode direct1 --stream-reserve $r; done
