# round-2 confirmation: smoke, the new GPU tests of this round
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke.log
timeout -k 10 ${BMI_T:-900} python -m pytest tests -m gpu -x -q --durations=12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_suite.log | tail -30
