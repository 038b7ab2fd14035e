#!/bin/bash
# round 3: the torus set (Bg 2^10, 48-bit key), plain + unrolled: parity, noise, inverses
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-900} python -m pytest tests/test_gpu_torus_unrolled.py tests/test_gpu_parity.py -m gpu -x -q -s -k "${BMI_K:-torus}" --durations=8 2>&1 | grep --line-buffered -v amdgpu.ids > gpurun_out/r3_torus.log
rc=$?
tail -40 gpurun_out/r3_torus.log
exit $rc
