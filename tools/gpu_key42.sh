set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "42_bits or torus64" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_key42.log | tail -8
BMI_BSK_PRECISION=42 timeout -k 10 200 python tools/br_timing.py 1,256,8192 0 65 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_key42.log
