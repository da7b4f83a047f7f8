#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes over bench.py, the bench lines of the
# BASELINE configurations (parity and CPU legs ON) and the latency / upload tools.  Everything lands under gpurun_out/final/;
# profiles/summarize.py turns it into the committed files.   usage: tools/collect_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# per-kernel times (trace only, no counters)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 --no-host-clouds > $O/kt_bench.json 2> $O/kt.log
# HBM-side traffic of the sweep: FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_sweep --output-format csv -d $O/fetch -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --steps 4 --warmup 1 > /dev/null 2> $O/fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_sweep --output-format csv -d $O/write -- python $R/bench.py --cpu-seconds 0 --no-host-clouds --steps 4 --warmup 1 > /dev/null 2> $O/write.log
rm -rf $O/kt/*/*.db $O/fetch/*/*.db $O/write/*/*.db
cd $R
# SQ / TCP / TCC counters of the sweep: the headline workload and the live nodelet's configuration
timeout 500 tools/pmc_kernel.sh k_sweep sq_direct7 > $O/pmc_sq_direct7.txt 2>&1
timeout 500 tools/pmc_kernel.sh k_sweep sq_pca_direct1 --variant pca --mode direct1 > $O/pmc_sq_pca_direct1.txt 2>&1
# bench lines (CPU baseline + pose-by-pose parity legs on)
timeout 600 python bench.py 2> $O/bench.log | tail -1 > $O/bench.json
timeout 400 python bench.py --variant pca --mode direct1 2>> $O/bench.log | tail -1 > $O/bench_pca_d1.json
timeout 400 python bench.py --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 2>> $O/bench.log | tail -1 > $O/bench_cfg5.json
timeout 400 python bench.py --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 2>> $O/bench.log | tail -1 > $O/bench_cfg5_d1.json
timeout 400 python bench.py --pairs 1536 --steps 5 --warmup 1 --cpu-seconds 30 2>> $O/bench.log | tail -1 > $O/bench_1536.json
timeout 400 python bench.py --total-pairs 4541 --steps 3 --warmup 1 --cpu-seconds 0 2>> $O/bench.log | tail -1 > $O/bench_cfg4_1gpu.json
# the N > 1 branch as two ranks sharing this GPU (gloo): functional evidence only, not a scaling number
LV_SLAM_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --total-pairs 542 --steps 3 --warmup 1 --cpu-seconds 0 2>> $O/bench.log | grep '^{' | tail -1 > $O/bench_2ranks_1gpu_gloo.json
timeout 200 python tools/latency_single.py 2>&1 | grep -v amdgpu.ids | tail -4 > $O/latency.txt
timeout 200 python tools/upload_rate.py 2>&1 | grep -v amdgpu.ids | tail -10 > $O/upload_rate.txt
ls -la $O
