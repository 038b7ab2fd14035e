#!/bin/bash
# the f64 accumulator of the torus wave-pair kernel: parity first, then timing
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "torus" > gpurun_out/r3_t64_accf.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r3_t64_accf.log | tail -30; exit 1; }
grep -v amdgpu.ids gpurun_out/r3_t64_accf.log | tail -3
timeout -k 5 200 python tools/br_timing.py 1,600,8192 0 65 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3_t64_accf.log
