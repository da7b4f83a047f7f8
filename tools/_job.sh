cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/fuzz
timeout 700 python tools/fuzz_parity.py --arith 1 --cases 100000 --seed 61 --seconds 300 --stream 0.15 2>&1 | grep -v amdgpu.ids | tail -25
