#!/bin/bash
# instruction-cache counters of the blind-rotation kernels (torus FFT pair kernel, 49-bit pair kernel)
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/icache
for qb in 65 49; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH_LEVEL --output-format csv -d gpurun_out/icache/q$qb -- python3 tools/br_timing.py 8192 0 $qb > gpurun_out/icache_$qb.log 2>&1 || echo "pmc failed $qb"
done
python3 - <<'PY' | tee gpurun_out/r3_icache.log
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/icache/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "blind_rotate" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k, {c: f"{x:.4g}" for c, x in v.items()})
PY
rm -rf gpurun_out/icache
