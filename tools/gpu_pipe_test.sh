#!/bin/bash
# the flag-synchronised unrolled latency kernel: one ciphertext first (a hang must not outlive its timeout), then timings A/B
mkdir -p gpurun_out
L=gpurun_out/r3_pipe.log
: > $L
BMI_UNROLL=2 timeout -k 5 90 python tools/br_timing.py 1 0 ${BMI_QB:-49} 2>&1 | grep -v amdgpu.ids >> $L || { echo "first launch failed or hung" >> $L; cat $L; exit 1; }
cat $L
grep -q '"bit_exact": true' $L || { echo "NOT bit exact"; exit 1; }
BMI_UNROLL=2 timeout -k 5 200 python tools/br_timing.py 1,64,256,512,8192 0 ${BMI_QB:-49} 2>&1 | grep -v amdgpu.ids >> $L &&
BMI_TFHE_LIB=$PWD/bounty-matrix-inversion_amd/lib/libbmi_tfhe_nopipe.so BMI_UNROLL=2 timeout -k 5 200 python tools/br_timing.py 1,64,256,512,8192 0 ${BMI_QB:-49} 2>&1 | grep -v amdgpu.ids >> $L
rc=$?
cat $L
exit $rc
