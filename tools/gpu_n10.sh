set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_inverse.py -m gpu -x -q -s -k "larger_sizes or 3x3_inverse_matches or 8x8" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_n10.log | grep -E "encrypted|passed|failed"
