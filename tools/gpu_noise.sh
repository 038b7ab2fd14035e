set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-600} python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "noise_matches" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_noise.log | tail -20
