#!/bin/bash
# One gpurun call's worth of work (GPU box).  usage: gpurun -- 'bash tools/gpu_job.sh <stage>...'; everything lands under gpurun_out/<stage>/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for stage in "$@"; do
  O=$R/gpurun_out/$stage; mkdir -p $O
  case $stage in
    tests)      timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    tests_new)  timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_gpu_parity.py::test_repeated_uploads_into_one_slot_land_in_call_order -x -q 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    bench)      (time timeout 900 python bench.py) > $O/bench.json 2> $O/bench.log; tail -c 3000 $O/bench.json; tail -5 $O/bench.log ;;
    bench20)    (time timeout 900 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.log; tail -c 3000 $O/bench.json; tail -5 $O/bench.log ;;
    pmc_cfg5d1) timeout 900 tools/pmc_kernel.sh k_sweep run_cfg5_d1 --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc.txt 2>&1; cat $O/pmc.txt ;;
    pmc_build)  timeout 900 tools/pmc_kernel.sh 'k_leafsum|k_rs_scatter|k_rs_hist|k_voxels|k_mark|k_minmax' run_build > $O/pmc.txt 2>&1; cat $O/pmc.txt ;;
    pmc_update) timeout 900 tools/pmc_kernel.sh 'k_update' run_update --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc.txt 2>&1; cat $O/pmc.txt ;;
    kstats_cfg5d1) timeout 600 tools/kstats.sh kstats_cfg5d1_run --no-host-clouds --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 --steps 10 --warmup 2 > $O/out.txt 2>&1; cat $O/out.txt ;;
    kstats)     timeout 600 tools/kstats.sh kstats_run --no-host-clouds --steps 20 --warmup 3 > $O/out.txt 2>&1; cat $O/out.txt ;;
    tests_seq)  timeout 1500 python -m pytest tests/test_sequence.py -x -q -s 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    latency)    (timeout 300 python tools/latency_single.py; LATENCY_MODE=1 timeout 300 python tools/latency_single.py; MI355NDT_FINE_TILES=1 LATENCY_MODE=1 timeout 300 python tools/latency_single.py) 2>&1 | grep -v amdgpu.ids > $O/latency.txt; cat $O/latency.txt ;;
    trace_cfg5d1) cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 --steps 3 --warmup 1 > $O/bench.json 2> $O/kt.log; cd $R
                python tools/dispatch_table.py $O/kt 'k_sweep|k_update' > $O/dispatches.txt; rm -rf $O/kt; tail -80 $O/dispatches.txt ;;
    trace_default) cd /tmp && export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --steps 3 --warmup 1 > $O/bench.json 2> $O/kt.log; cd $R
                python tools/dispatch_table.py $O/kt 'k_' > $O/dispatches.txt; rm -rf $O/kt; tail -80 $O/dispatches.txt ;;
    prefiltered) timeout 600 python bench.py --prefiltered --azimuth 2048 --pairs 64 --variant pca --mode direct1 --cpu-seconds 10 > $O/bench.json 2> $O/bench.log; tail -c 2500 $O/bench.json ;;
    tests_bench) timeout 1500 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    seq)        timeout 300 python tools/seq_run.py 129 2>&1 | grep -v amdgpu.ids > $O/seq.txt; cat $O/seq.txt
                cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/seq_run.py 65 > $O/seq_traced.txt 2> $O/kt.log; cd $R
                python tools/seq_kernels.py $O/kt > $O/seq_kernels.txt; rm -rf $O/kt; cat $O/seq_kernels.txt ;;
    tests_r4)   timeout 2400 python -m pytest tests/test_reference_golden.py tests/test_adaptor.py tests/test_gpu_parity.py -m gpu -x -q -k "reference or adaptor or calculate_score or convert_transform or f32_sum_order" 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    tests_async) timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "one_launch or dealt" 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    bench_q)    (time timeout 900 python bench.py --cpu-seconds 8 --no-host-clouds --seq-frames 0 --config4-pairs 0) > $O/bench.json 2> $O/bench.log; tail -c 3000 $O/bench.json; tail -5 $O/bench.log ;;
    bench_q0)   (time MI355NDT_ASYNC=0 timeout 900 python bench.py --cpu-seconds 8 --no-host-clouds --seq-frames 0 --config4-pairs 0) > $O/bench.json 2> $O/bench.log; tail -c 3000 $O/bench.json; tail -5 $O/bench.log ;;
    timeline)   (for a in 0 1; do MI355NDT_ASYNC=$a MAXIT=64 MI355NDT_LIB=$R/lv_slam_amd/libexp_tl.so timeout 300 python tools/sweep_timeline.py; done) 2>&1 | grep -v amdgpu.ids > $O/tl.txt; cat $O/tl.txt ;;
    fuzz_async) MI355NDT_ASYNC=2 timeout 900 python tools/fuzz_parity.py --cases 100000 --seed 41 --seconds 420 > $O/fuzz.txt 2>&1; tail -15 $O/fuzz.txt ;;
    bench_d1)   for c in "--variant pca --mode direct1" "--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128" ""; do timeout 600 python bench.py $c --cpu-seconds 6 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"][:60], d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"parity\"][\"pairs_checked\"], d[\"parity\"][\"iterations_equal\"])"; done > $O/out.txt 2>&1; cat $O/out.txt ;;
    timeline_d1) (MODE=direct1 VARIANT=pca MAXIT=64 MI355NDT_LIB=$R/lv_slam_amd/libexp_tl.so timeout 300 python tools/sweep_timeline.py; MODE=direct1 VARIANT=pca RESOLUTION=0.5 AZIMUTH=2048 PAIRS=128 MAXIT=64 MI355NDT_LIB=$R/lv_slam_amd/libexp_tl.so timeout 300 python tools/sweep_timeline.py) 2>&1 | grep -v amdgpu.ids > $O/tl.txt; cat $O/tl.txt ;;
    tests_stream) timeout 1800 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -30 > $O/pytest.txt; cat $O/pytest.txt ;;
    tests_async_all) timeout 1800 python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt ;;
    kstats_leaf) for v in 0 1; do MI355NDT_LEAF_SORTED=$v timeout 600 tools/kstats.sh kstats_leaf$v --no-host-clouds --no-stream --steps 20 --warmup 3 > $O/out$v.txt 2>&1; grep -E "k_leafsum|k_sorted|k_mark|k_rs_|k_voxels|k_minmax|k_rank|k_align" $O/out$v.txt; done
                 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "voxel or grid" 2>&1 | tail -5 > $O/pytest0.txt; cat $O/pytest0.txt
                 MI355NDT_LEAF_SORTED=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "voxel or grid" 2>&1 | tail -5 > $O/pytest1.txt; cat $O/pytest1.txt ;;
    reserve_sweep) for r in 0 32 64 128; do for c in "" "--variant pca --mode direct1" "--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128" "--variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128"; do timeout 600 python bench.py $c --stream-reserve $r --stream-contexts ${NCTX:-4} --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('reserve $r', d['config']['workload'][:44], 'stream', d['value'], 'sync', d['value_synchronous'], 'ms', d['ms_per_step'], d['ms_per_step_synchronous'], 'launch', r['avg_launch_us'], 'build', r['build_ms_per_step'], 'frac', r['frac'], d['config']['stream']['pairs_handed_over'])"; done; done > $O/out.txt 2>&1; cat $O/out.txt ;;
    ab_r04) for rep in 1 2; do for c in "" "--variant pca --mode direct1" "--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128"; do
                 (cd tools/ab_r04 && timeout 600 python bench.py $c --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('r04 ', d['config']['workload'][:40], d['value'], d['ms_per_step'], 'launch', r['avg_launch_us'], 'build', r['build_ms_per_step'])")
                 timeout 600 python bench.py $c --no-stream --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('HEAD', d['config']['workload'][:40], d['value'], d['ms_per_step'], 'launch', r['avg_launch_us'], 'build', r['build_ms_per_step'])"
               done; done > $O/out.txt 2>&1; cat $O/out.txt ;;
    thresh_sweep) for t in 4 8 16 32 64; do for c in "" "--variant pca --mode direct1" "--variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128" "--variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128"; do MI355NDT_STREAM_THRESH=$t timeout 600 python bench.py $c --cpu-seconds 0 --no-host-clouds --seq-frames 0 --config4-pairs 0 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('thresh $t', d['config']['workload'][:44], 'stream', d['value'], 'sync', d['value_synchronous'], 'launch', r['avg_launch_us'], 'frac', r['frac'], d['config']['stream']['pairs_handed_over'], d['config']['stream']['launches'])"; done; done > $O/out.txt 2>&1; cat $O/out.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
