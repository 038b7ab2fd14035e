#!/bin/bash
# round 3: random matrices through the encrypted inverse (CSPRNG keys) with the round-3 circuits: 49-bit unrolled / plain, 2^64 torus unrolled
mkdir -p gpurun_out
L=gpurun_out/r3_random_inverses.log
: > $L
run() { timeout -k 10 400 python tools/gpu_random_inverses.py "$@" 2>&1 | grep --line-buffered -v amdgpu.ids >> $L; }
run 2 400 49 2 && run 3 150 49 2 && run 4 40 49 2 && run 3 60 49 1 && run 2 300 65 2 && run 3 100 65 2 && run 4 25 65 2
rc=$?
cat $L
exit $rc
