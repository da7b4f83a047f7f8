run() { timeout 600 python bench.py "$@" --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('tolerance_mode') or {}
print('$*', '| stream', d['value'], 'sync', d['value_synchronous'], 'ms', d['ms_per_step'], 'launch', r['avg_launch_us'], '| tol stream', t.get('value_tolerance_mode_streamed'), 'sync', t.get('value_tolerance_mode_synchronous'))"; }
for r in 0 32 64 96 128; do run --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 --stream-reserve $r; done
for r in 0 64 128; do run --variant pca --mode direct7 --stream-reserve $r; done
for r in 64 96 128; do run --stream-reserve $r; done
for r in 96 128 160; do run --variant pca --mode direct1 --stream-reserve $r; done
for r in 96 128; do run --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 --stream-reserve $r; done
