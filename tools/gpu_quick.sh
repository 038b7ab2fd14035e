set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-500} python tools/br_timing.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quick.log
