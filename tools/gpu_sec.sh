set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-700} python -m pytest tests/test_gpu_parity.py tests/test_gpu_inverse.py -m gpu -x -q -s -k "secure128" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_sec.log | tail -20
