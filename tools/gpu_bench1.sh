set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --batch 4096 --cpu-seconds 8 2>&1 | tee gpurun_out/bench1.log | tail -5
