#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_ks_ab2.log
: > $L
for qb in 49 65 64; do
  timeout -k 5 150 python tools/br_timing.py 1,256,8192 0 $qb 2>&1 | grep -v amdgpu.ids | cut -c1-150 >> $L || { echo FAILED >> $L; cat $L; exit 1; }
done
cat $L
