set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-800} python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torus64" --durations=8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_t64.log | tail -40
