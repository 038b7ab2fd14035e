#!/bin/bash
# BASELINE config 5 (8x8, len 48, ints 16) on the 2^64 torus, opt-in (tests/test_gpu_inverse.py): plain torus kernels
mkdir -p gpurun_out
BMI_TEST_TORUS_8X8=1 PYTHONUNBUFFERED=1 timeout -k 10 900 python -u -m pytest tests/test_gpu_inverse.py -m gpu -x -q -s -k "8x8 and (torus64 or p49)" > gpurun_out/r3_torus_8x8.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/r3_torus_8x8.log | tail -8
exit $rc
