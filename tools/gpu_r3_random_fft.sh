#!/bin/bash
# round 3: random matrices through the encrypted inverse (CSPRNG keys) on the 2^64 torus with the PLAIN key: every look-up runs on the
# kernels whose exact limb products go through the f64 FFT (latency form for narrow levels, wave pairs for wide ones)
mkdir -p gpurun_out
L=gpurun_out/r3_random_inverses_fft.log
: > $L
run() { timeout -k 10 500 python tools/gpu_random_inverses.py "$@" 2>&1 | grep --line-buffered -v amdgpu.ids >> $L; }
run 2 300 65 1 && run 3 100 65 1 && run 4 25 65 1
rc=$?
cat $L
exit $rc
