# random matrices through the encrypted inverse with the UNROLLED bootstrap key (CSPRNG keys, key noise 2^-41)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 300 python tools/gpu_random_inverses.py 2 600 49 2 && timeout -k 10 300 python tools/gpu_random_inverses.py 3 120 49 2 && timeout -k 10 300 python tools/gpu_random_inverses.py 4 40 49 2) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/random_inverses_unrolled.log
