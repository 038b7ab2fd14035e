set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 300 python tools/gpu_random_inverses.py 2 400 49 && timeout -k 10 300 python tools/gpu_random_inverses.py 3 60 49 && timeout -k 10 200 python tools/gpu_random_inverses.py 4 12 49 && timeout -k 10 200 python tools/gpu_random_inverses.py 2 100 65) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/random_inverses.log
