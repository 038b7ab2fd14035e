# full GPU test suite, then the default bench line (one gpurun call)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-1000} python -m pytest tests -m gpu -x -q --durations=15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_suite.log | tail -40
