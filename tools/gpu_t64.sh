set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-800} python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torus64" --durations=5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_t64.log | tail -12
timeout -k 10 200 python tools/br_timing.py 1,256,512 0 65 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_t64.log
BMIP_BS_LEVELS=2 timeout -k 10 200 python tools/br_timing.py 256 0 65 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_t64.log
