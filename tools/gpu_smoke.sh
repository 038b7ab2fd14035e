set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke.log | tail -5
timeout -k 10 600 python bench.py --inverse-sizes 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_quick.log | tail -2 | cut -c1-3000
