#!/bin/bash
# torus kernels through the floating-point transform: parity first, then timing against the exact-transform kernels
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torus and (variant or known_answer or bit_exact)" > gpurun_out/r3_t64f.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r3_t64f.log | tail -30; exit 1; }
grep -v amdgpu.ids gpurun_out/r3_t64f.log | tail -3
timeout -k 5 300 python tools/br_timing.py ${BMI_DEV_BATCHES:-1,256,600} ${BMI_DEV_VARIANTS:-6,4,5} 65 2>&1 | grep --line-buffered -v amdgpu.ids | cut -c1-210 | tee -a gpurun_out/r3_t64f.log
