#!/bin/bash
# One parameterised runner for everything that is executed on the GPU box (replaces the per-experiment gpu_*.sh scripts of rounds 1-3).
#   gpurun --timeout 900 -- 'bash tools/gpu.sh <recipe> [args]'
# Every recipe writes under gpurun_out/ (scratch, merged back by gpurun); copy what is to be judged into profiles/.
#
#   suite [pytest args]          the whole GPU suite as the driver runs it, with durations           -> gpurun_out/gpu_suite.log
#   test  <pytest args>          some tests, output to a file (a silent run is taken to be hung)       -> gpurun_out/gpu_test.log
#   final [tag]                  the driver's exact bench command + rocprofv3 kernel trace + three PMC passes of it, summarised per
#                                kernel instantiation and launch shape (tools/summarize_prof.py)      -> gpurun_out/final/
#   ab    "<lib suffixes>" <tool and args>   the same timing tool against several builds of the library (make variant NAME=x ->
#                                libbmi_tfhe_x.so; "" = the product build), e.g.  ab "'' _kd3" tools/preset_timing.py secure128_torus 1,256
#   timing <br_timing args>      tools/br_timing.py (batches, kernel variants, q_bits)
#   preset <preset> [batches]    tools/preset_timing.py
#   phase  <t64f|t64w> [batches] per-phase cycles of the debug build (make -C csrc prof)
#   inverse [sizes]              encrypted-inverse wall-clocks on the default engine (bmi_amd/inverse_bench.py)
#   smoke                        __graft_entry__.smoke()
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
L=bounty-matrix-inversion_amd/lib
recipe=$1; shift
case "$recipe" in
  suite)
    timeout -k 10 ${BMI_T:-1150} python -u -m pytest tests/ -x -q -m gpu --durations=25 "$@" > gpurun_out/gpu_suite.log 2>&1; rc=$?
    grep -v amdgpu.ids gpurun_out/gpu_suite.log | tail -45; exit $rc ;;
  test)
    timeout -k 10 ${BMI_T:-850} python -u -m pytest -x -q -s "$@" > gpurun_out/gpu_test.log 2>&1; rc=$?
    grep -v "amdgpu.ids\|^$" gpurun_out/gpu_test.log | tail -40; exit $rc ;;
  ab)
    libs=$1; shift
    : > gpurun_out/ab.log
    for round in 1 ${BMI_AB_ROUNDS:+2}; do for v in $libs; do
      [ "$v" = "''" ] && v=""
      echo "== libbmi_tfhe$v.so" | tee -a gpurun_out/ab.log
      BMI_TFHE_LIB=$PWD/$L/libbmi_tfhe$v.so timeout -k 10 ${BMI_T:-200} python "$@" 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a gpurun_out/ab.log || exit 1
    done; done ;;
  timing) timeout -k 10 ${BMI_T:-500} python tools/br_timing.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/timing.log ;;
  preset) timeout -k 10 ${BMI_T:-300} python tools/preset_timing.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/preset_timing.log ;;
  phase)
    k=$1; shift
    timeout -k 10 ${BMI_T:-300} python tools/phase_prof_$k.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phase_$k.log ;;
  inverse)
    timeout -k 10 ${BMI_T:-900} python - "${1:-2,3,4}" <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/inverse.log
import sys, json, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "bounty-matrix-inversion_amd"))
from bmi_amd import tfhe, inverse_bench
eng = tfhe.Engine(); eng.keygen(0x5EED)
print(json.dumps(inverse_bench.run(eng, tuple(int(x) for x in sys.argv[1].split(","))), indent=1))
PY
    ;;
  smoke) timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke.log ;;
  final)
    TAG=${1:-r04}; B=8192; F=gpurun_out/final
    rm -rf $F; mkdir -p $F/summary
    # the driver's exact command, on a cold program cache as on the driver's box
    CMD="--steps 20 --warmup 5"
    echo "bench ($CMD)" > $F/progress.log; T0=$(date +%s)
    BMI_CACHE_DIR=/tmp/bmi_cold_cache_$$ python bench.py $CMD > $F/bench_default.json 2> $F/bench_default.err || { tail -5 $F/bench_default.err; exit 1; }
    cp gpurun_out/bench_details.json $F/bench_details.json 2>/dev/null
    echo "bench took $(( $(date +%s) - T0 )) s" >> $F/progress.log
    # the same timed region under the profiler (legs, CPU baseline and inverses off: they are not the timed region)
    PROF="python3 bench.py $CMD --batch $B --no-cpu-baseline --no-inverse --no-second-field"
    echo trace >> $F/progress.log
    rocprofv3 --kernel-trace --stats --output-format csv -d $F/prof/trace -- $PROF > $F/prof_trace.log 2>&1 || { tail -5 $F/prof_trace.log; exit 1; }
    grep -o '"sclk_mhz": [0-9.]*' $F/prof_trace.log | head -1 > $F/summary/sclk_during_trace.txt
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
      set -- $pass; name=$1; shift
      echo "pmc $name" >> $F/progress.log
      rocprofv3 --pmc "$@" --output-format csv -d $F/prof/$name -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse --no-second-field > $F/prof_$name.log 2>&1 || echo "$name pass failed"
    done
    python3 tools/summarize_prof.py $TAG $B $F/prof $F/summary $F/prof_trace.log > $F/summary/summary.log 2>&1 || { tail -20 $F/summary/summary.log; exit 1; }
    rm -rf $F/prof
    tail -c 1500 $F/bench_default.json; echo; cat $F/summary/summary.log ;;
  *) sed -n 2,20p "$0"; exit 2 ;;
esac
