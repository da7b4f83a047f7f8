#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes over bench.py, and the bench lines of the
# other BASELINE configurations.  Everything lands under gpurun_out/; profiles/summarize.py turns it into the committed files.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --cpu-seconds 0 > $O/kt_bench.json 2> $O/kt.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_sweep --output-format csv -d $O/fetch -- python $R/bench.py --cpu-seconds 0 --steps 4 --warmup 1 > /dev/null 2> $O/fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_sweep --output-format csv -d $O/write -- python $R/bench.py --cpu-seconds 0 --steps 4 --warmup 1 > /dev/null 2> $O/write.log
cd $R
timeout 600 python bench.py 2> $O/bench.log | tail -1 > $O/bench.json
timeout 300 python bench.py --cpu-seconds 0 --variant pca --mode direct1 2>> $O/bench.log | tail -1 > $O/bench_pca_d1.json
timeout 300 python bench.py --cpu-seconds 0 --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 2>> $O/bench.log | tail -1 > $O/bench_cfg5.json
timeout 300 python bench.py --cpu-seconds 0 --pairs 1536 --steps 5 --warmup 1 2>> $O/bench.log | tail -1 > $O/bench_1536.json
timeout 200 python tools/latency_single.py 2>&1 | tail -2 > $O/latency.txt
ls -la $O
