set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_c_abi.py tests/test_gpu_inverse.py -m gpu -x -q -k "c_program or two_rank or random_matrices" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_misc.log | tail -12
