cd $GRAFT_REPO_ROOT
for c in "--variant pca --mode direct1" "--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128" "--variant omp --mode direct1"; do for ar in 0 1; do python bench.py $c --arith $ar --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs --no-tolerance-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('auto arith $ar', d['config']['workload'][:34], 'stream', d['value_streamed'], 'sync', d['value_synchronous'], 'launch', r['avg_launch_us'], 'build', r['build_ms_per_step'], d['config']['stream'])"; done; done
timeout 900 python -m pytest tests/test_stream_gpu.py -q -x 2>&1 | tail -3
