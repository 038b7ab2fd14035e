set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_parameter_shape" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_lb.log | tail -15
BMIP_BS_LEVELS=2 timeout -k 10 200 python tools/br_timing.py 256,8192 0 49 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_lb.log
BMIP_BS_LEVELS=1 BMIP_BS_BASE_LOG=23 timeout -k 10 200 python tools/br_timing.py 256,8192 0 49 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_lb.log
BMIP_BS_LEVELS=2 timeout -k 10 200 python tools/br_timing.py 8192 0 65 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gpu_lb.log
