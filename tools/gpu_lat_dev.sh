#!/bin/bash
# the torus latency kernel on half transforms: stand-alone transform check, parity, timing
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/fft_half_check tests/hip/fft_half_check.hip 2>/dev/null && timeout -k 5 60 /tmp/fft_half_check || exit 1
PYTHONUNBUFFERED=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_torus_fft.py -m gpu -x -q -k "torus and (variant or bit_exact_every or known_answer)" > gpurun_out/r3_lat_dev.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r3_lat_dev.log | tail -30; exit 1; }
grep -v amdgpu.ids gpurun_out/r3_lat_dev.log | tail -2
timeout -k 5 300 python tools/br_timing.py 1,256 6,4 65 2>&1 | grep --line-buffered -v amdgpu.ids | cut -c1-120 | tee -a gpurun_out/r3_lat_dev.log
