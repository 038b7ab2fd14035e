# end-of-iteration measurement: default bench line (+ inverse wall-clock), then rocprofv3 trace + PMC passes at the same batch
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
python bench.py --inverse-sizes ${BMI_INV_SIZES:-2,3,4} > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python bench.py --q-bits 65 --no-cpu-baseline --inverse-sizes 2 > gpurun_out/final/bench_torus64.json 2> gpurun_out/final/bench_torus64.err
tail -c 3000 gpurun_out/final/bench_default.json
B=8192
rm -rf gpurun_out/final/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof/trace -- python3 bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/final/prof_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/prof/fetch -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/final/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/prof/write -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/final/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/final/prof/sq -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse > gpurun_out/final/prof_sq.log 2>&1 || echo "sq pass failed"
echo done
