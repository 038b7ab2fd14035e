set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_inverse.py -m gpu -x -q --durations=5 2>&1 | tee gpurun_out/gpu_inverse.log | tail -30
