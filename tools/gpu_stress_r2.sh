# round-2 kernels under the stress tool: torus (both kernels), the (2, 2^15) / (1, 2^23) templates, N = 2048 at l = 2
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 200 python tools/gpu_stress.py 65 90 && BMIP_BS_LEVELS=2 timeout -k 10 120 python tools/gpu_stress.py 65 45 && \
 BMIP_BS_LEVELS=2 timeout -k 10 120 python tools/gpu_stress.py 49 45 && BMIP_BS_LEVELS=1 BMIP_BS_BASE_LOG=23 timeout -k 10 120 python tools/gpu_stress.py 49 45 && \
 BMIP_BS_LEVELS=2 timeout -k 10 120 python tools/gpu_stress.py 49 40 11 && timeout -k 10 120 python tools/gpu_stress.py 49 45) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stress_r2.log | grep -E "stress ok|Error|assert" 
