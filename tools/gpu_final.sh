#!/bin/bash
# end-of-iteration measurement: the default bench line (all legs + inverse wall-clocks), then rocprofv3 kernel trace + three PMC
# passes of the same command at the same batch, summarised per kernel instantiation and launch shape (tools/summarize_prof.py)
# into gpurun_out/final/summary/ - copy what is to be judged into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${BMI_TAG:-r03}
B=8192
mkdir -p gpurun_out/final/summary
echo "bench (default)" > gpurun_out/final/progress.log
T0=$(date +%s)
# a cold program cache, as on the driver's box: the default bench line must still finish within minutes
BMI_CACHE_DIR=/tmp/bmi_cold_cache_$$ python bench.py --inverse-sizes ${BMI_INV_SIZES:-2,3,4} > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || { tail -5 gpurun_out/final/bench_default.err; exit 1; }
echo "bench (default, cold program cache) took $(( $(date +%s) - T0 )) s" >> gpurun_out/final/progress.log
echo "bench (49-bit field as the headline)" >> gpurun_out/final/progress.log
python bench.py --q-bits 49 --no-inverse --cpu-seconds 8 > gpurun_out/final/bench_p49.json 2> gpurun_out/final/bench_p49.err || { tail -5 gpurun_out/final/bench_p49.err; exit 1; }
rm -rf gpurun_out/final/prof
PROF="python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-inverse"
echo "trace" >> gpurun_out/final/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof/trace -- $PROF > gpurun_out/final/prof_trace.log 2>&1 || { tail -5 gpurun_out/final/prof_trace.log; exit 1; }
echo "pmc fetch" >> gpurun_out/final/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/prof/fetch -- $PROF > gpurun_out/final/prof_fetch.log 2>&1 || echo "fetch pass failed"
echo "pmc write" >> gpurun_out/final/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/prof/write -- $PROF > gpurun_out/final/prof_write.log 2>&1 || echo "write pass failed"
echo "pmc sq" >> gpurun_out/final/progress.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/final/prof/sq -- $PROF > gpurun_out/final/prof_sq.log 2>&1 || echo "sq pass failed"
echo "summarise" >> gpurun_out/final/progress.log
python3 tools/summarize_prof.py $TAG $B gpurun_out/final/prof gpurun_out/final/summary > gpurun_out/final/summary/summary.log 2>&1 || { tail -20 gpurun_out/final/summary/summary.log; exit 1; }
# keep the heads of the raw files for debugging the summariser, not the bulk
for f in $(find gpurun_out/final/prof -name "*.csv"); do head -3 $f > gpurun_out/final/summary/head_$(basename $(dirname $(dirname $f)))_$(basename $f).txt; done
rm -rf gpurun_out/final/prof
tail -c 2500 gpurun_out/final/bench_default.json
echo
cat gpurun_out/final/summary/summary.log
echo done
