#!/bin/bash
# A/B of the flag-synchronised unrolled latency kernel (priorities, wake-up form) against the barrier form, one call
mkdir -p gpurun_out
L=gpurun_out/r3_pipe_ab2.log
: > $L
for lib in ${BMI_LIBS:-libbmi_tfhe_pw0.so libbmi_tfhe_pw1.so libbmi_tfhe_pw2.so libbmi_tfhe.so}; do
  echo "== $lib" >> $L
  BMI_TFHE_LIB=$PWD/bounty-matrix-inversion_amd/lib/$lib BMI_UNROLL=2 timeout -k 5 120 python tools/br_timing.py 1,256,8192 0 49 2>&1 | grep -v amdgpu.ids | cut -c1-60 >> $L || { echo FAILED >> $L; cat $L; exit 1; }
done
cat $L
