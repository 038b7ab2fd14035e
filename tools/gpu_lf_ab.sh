#!/bin/bash
mkdir -p gpurun_out
LOG=gpurun_out/r3_lf_ab.log
: > $LOG
for lib in ${BMI_AB_LIBS:-libbmi_tfhe.so}; do
  echo "== $lib" | tee -a $LOG
  BMI_TFHE_LIB=$PWD/bounty-matrix-inversion_amd/lib/$lib timeout -k 5 200 python tools/br_timing.py 1,256 6 65 2>&1 | grep --line-buffered -v amdgpu.ids | cut -c1-60 | tee -a $LOG
done
