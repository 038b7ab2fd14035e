# rocprofv3 passes for the bench workload: kernel trace + stats, then HBM read / write counters (separate passes)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${1:-4096}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/prof/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/prof/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/prof/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof/sq -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/prof/bench_sq.log 2>&1 || echo "sq pass failed"
find gpurun_out/prof -name "*.csv" | head -40
