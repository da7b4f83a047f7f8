#!/bin/bash
# builds lv_slam_amd/libvar_<name>.so with extra -D flags (kernel tuning variants for same-box A/B runs through MI355NDT_LIB): tools/build_variant.sh <name> "<flags>"
# (the ORD = 1 unit is taken from the regular build: it does not depend on the tolerance-arithmetic knobs)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -Wno-unused-value -I$R/include -I$R/lv_slam_amd/csrc $*"
D=/tmp/var_$N; mkdir -p $D
/opt/rocm/bin/hipcc $FL -c $R/lv_slam_amd/csrc/mi355_ndt.hip -o $D/a.o &
/opt/rocm/bin/hipcc $FL -Rpass-analysis=kernel-resource-usage -c $R/lv_slam_amd/csrc/mi355_ndt_fast.hip -o $D/c.o 2> $D/fast.log &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/a.o $R/lv_slam_amd/csrc/mi355_ndt_ord1.hip.o $D/c.o -o $R/lv_slam_amd/libvar_$N.so
python3 $R/tools/kres.py $D/fast.log 'k_align_async|k_sweep<.*8, false' 
