#!/bin/bash
# generic same-box A/B of library builds: BMI_LIBS="a.so b.so", BMI_ARGS="<br_timing args>", optional BMI_UNROLL
mkdir -p gpurun_out
L=gpurun_out/r3_ab_generic.log
: > $L
for rep in 1 2; do for lib in $BMI_LIBS; do
  echo "== $lib" >> $L
  BMI_TFHE_LIB=$PWD/bounty-matrix-inversion_amd/lib/$lib timeout -k 5 150 python tools/br_timing.py $BMI_ARGS 2>&1 | grep -v amdgpu.ids | cut -c1-70 >> $L || { echo FAILED >> $L; cat $L; exit 1; }
done; done
cat $L
