#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes over bench.py, the bench lines of the
# BASELINE configurations (parity and CPU legs ON), the latency-mode tools.  Everything lands under gpurun_out/final/;
# profiles/summarize.py turns it into the committed files.   usage: tools/collect_profiles.sh [part ...]   (default: all parts)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
PARTS=${@:-"trace pmc tol bench extra"}
QUIET="--cpu-seconds 0 --no-host-clouds --config4-pairs 0 --seq-frames 0 --no-other-configs"
# FETCH_SIZE and WRITE_SIZE of the sweep launches in separate passes, no trace domains (--no-stream: the synchronous job's launches only, one kind of
# dispatch per pass; the streamed job runs the same kernel on the same pairs)
traffic_pass() {   # <tag> <arith> <bench args...>
  local tag=$1 ar=$2; shift; shift
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/fetch_$tag $O/write_$tag
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'k_sweep|k_align_async' --output-format csv -d $O/fetch_$tag -- python $R/bench.py --arith $ar $QUIET --no-tolerance-mode --no-stream --steps 4 --warmup 1 "$@" > $O/traffic_bench_$tag.json 2> $O/fetch_$tag.log
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'k_sweep|k_align_async' --output-format csv -d $O/write_$tag -- python $R/bench.py --arith $ar $QUIET --no-tolerance-mode --no-stream --steps 4 --warmup 1 "$@" > /dev/null 2> $O/write_$tag.log
  rm -rf $O/fetch_$tag/*/*.db $O/write_$tag/*/*.db
  cd $R
}
traffic_all() {    # <arith> <suffix>
  traffic_pass cfg3$2 $1
  traffic_pass pca_direct1$2 $1 --variant pca --mode direct1
  traffic_pass cfg5_d7$2 $1 --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128
  traffic_pass cfg5_d1$2 $1 --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128
}
for part in $PARTS; do
case $part in
trace)
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/kt
  # per-kernel times (trace only, no counters)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py $QUIET --no-tolerance-mode --steps 20 --warmup 3 > $O/kt_bench.json 2> $O/kt.log
  # HBM-side traffic of the sweep, EVERY timed configuration, both arithmetics: FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains
  # (--no-stream: the synchronous job's launches only, one kind of dispatch per pass; the streamed job runs the same kernel on the same pairs)
  traffic_all 0 ""
  cd /tmp
  rm -rf $O/kt/*/*.db
  cd $R
  timeout 400 tools/kstats.sh final_kd1 --no-host-clouds --variant pca --mode direct1 --steps 20 --warmup 3 > $O/kstats_pca_d1.txt 2>&1
  timeout 400 tools/kstats.sh final_kc5 --no-host-clouds --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 --steps 10 --warmup 2 > $O/kstats_cfg5_d1.txt 2>&1
  ;;
pmc)
  cd $R
  # SQ / TCP / TCC counters: the sweep on the headline workload, the live nodelet's configuration and config 5 with DIRECT1; the build kernels; the update
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_direct7 --no-stream --no-tolerance-mode > $O/pmc_sq_direct7.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_pca_direct1 --no-stream --no-tolerance-mode --variant pca --mode direct1 > $O/pmc_sq_pca_direct1.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_cfg5_d1 --no-stream --no-tolerance-mode --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc_sq_cfg5_d1.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_cfg5_d7 --no-stream --no-tolerance-mode --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc_sq_cfg5_d7.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_leafsum|k_rs_|k_voxels|k_mark|k_rank|k_minmax|k_griddesc|k_word_offsets' sq_build --no-stream --no-tolerance-mode > $O/pmc_build.txt 2>&1
  MI355NDT_LEAF_SORTED=1 timeout 500 tools/pmc_kernel.sh 'k_leafsum|k_sorted_points' sq_build_sorted --no-stream --no-tolerance-mode > $O/pmc_build_sorted.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_seq_update|k_update' sq_update --no-stream --no-tolerance-mode --seq-frames 65 > $O/pmc_update.txt 2>&1
  timeout 300 tools/pmc_calib.sh > $O/pmc_valu_calib.log 2>&1; cp $R/gpurun_out/pmc_calib/calib.json $O/pmc_valu_calib.json
  ;;
tol)
  # the tolerance arithmetic (bench.py --arith 1): kernel times, HBM traffic and SQ / TCP / TCC counters of the same four configurations, its build
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_tol -- python $R/bench.py --arith 1 $QUIET --no-tolerance-mode --steps 20 --warmup 3 > $O/kt_tol_bench.json 2> $O/kt_tol.log
  find $O/kt_tol -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats_tol.csv; rm -rf $O/kt_tol
  cd $R
  traffic_all 1 _tol
  # the same four configurations in the tolerance arithmetic (MI355NDT_ARITH=1: every engine of the run)
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_tol_direct7 --arith 1 --no-stream --no-tolerance-mode > $O/pmc_sq_tol_direct7.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_tol_pca_direct1 --arith 1 --no-stream --no-tolerance-mode --variant pca --mode direct1 > $O/pmc_sq_tol_pca_direct1.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_tol_cfg5_d1 --arith 1 --no-stream --no-tolerance-mode --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc_sq_tol_cfg5_d1.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_sweep|k_align_async' sq_tol_cfg5_d7 --arith 1 --no-stream --no-tolerance-mode --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 > $O/pmc_sq_tol_cfg5_d7.txt 2>&1
  timeout 500 tools/pmc_kernel.sh 'k_leafsum|k_rs_|k_voxels|k_mark|k_rank|k_minmax|k_griddesc|k_word_offsets' sq_tol_build --arith 1 --no-stream --no-tolerance-mode > $O/pmc_tol_build.txt 2>&1

  ;;
bench)
  cd $R
  # bench lines (CPU baseline + pose-by-pose parity legs on); the default line carries the config-4 block and the latency-mode leg
  timeout 900 python bench.py 2> $O/bench.log | tail -1 > $O/bench.json
  timeout 400 python bench.py --variant pca --mode direct1 --config4-pairs 0 --seq-frames 0 2>> $O/bench.log | tail -1 > $O/bench_pca_d1.json
  timeout 400 python bench.py --variant pca --mode direct7 --resolution 0.5 --azimuth 2048 --pairs 128 --config4-pairs 0 --seq-frames 0 2>> $O/bench.log | tail -1 > $O/bench_cfg5.json
  timeout 400 python bench.py --variant pca --mode direct1 --resolution 0.5 --azimuth 2048 --pairs 128 --config4-pairs 0 --seq-frames 0 2>> $O/bench.log | tail -1 > $O/bench_cfg5_d1.json
  timeout 400 python bench.py --pairs 1536 --steps 5 --warmup 1 --cpu-seconds 30 --config4-pairs 0 --seq-frames 0 --no-other-configs 2>> $O/bench.log | tail -1 > $O/bench_1536.json
  # BASELINE config 4's whole job on one GPU, through the RCCL gather (one forced rank), parity sample spread over pairs 0..4540
  LV_SLAM_BENCH_FORCE_DIST=1 MASTER_PORT=29541 timeout 600 python bench.py --total-pairs 4541 --steps 5 --warmup 1 --cpu-seconds 16 --no-host-clouds --seq-frames 0 --no-other-configs 2>> $O/bench.log | grep '^{' | tail -1 > $O/bench_cfg4_1gpu.json
  # the N > 1 branch as two ranks sharing this GPU (gloo), DEFAULT flags = what the driver's scaling runs pass: weak line + config-4 block + gather times
  LV_SLAM_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 5 --warmup 2 2>> $O/bench.log | grep '^{' | tail -1 > $O/bench_2ranks_1gpu_gloo.json
  timeout 600 python bench.py --prefiltered --azimuth 2048 --pairs 64 --variant pca --mode direct1 --cpu-seconds 10 2>> $O/bench.log | tail -1 > $O/bench_prefiltered.json
  ;;
extra)
  cd $R
  (timeout 200 python tools/latency_single.py; LATENCY_MODE=1 timeout 200 python tools/latency_single.py) 2>&1 | grep -v amdgpu.ids > $O/latency.txt
  timeout 300 python tools/seq_run.py 271 2>&1 | grep -v amdgpu.ids > $O/sequence.txt
  cd /tmp && export TMPDIR=/tmp; rm -rf $O/seqkt
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/seqkt -- python $R/tools/seq_run.py 65 > /dev/null 2> $O/seqkt.log; cd $R
  python tools/seq_kernels.py $O/seqkt >> $O/sequence.txt 2>&1; rm -rf $O/seqkt
  timeout 200 python tools/upload_rate.py 2>&1 | grep -v amdgpu.ids | tail -10 > $O/upload_rate.txt
  ;;
esac
done
# reduce on the box (the raw traces are far too large to travel back) and drop the raw directories
cd $R
python profiles/summarize.py r06 $O/reduced > $O/summarize.log 2>&1
rm -rf $O/kt $O/fetch_* $O/write_* $R/gpurun_out/final_kd1/kt $R/gpurun_out/final_kc5/kt $R/gpurun_out/pmc_sq_*
tail -5 $O/summarize.log
for f in bench.log kt.log; do echo "== $f"; tail -4 $O/$f; done
ls -la $O $O/reduced; du -sh $R/gpurun_out
