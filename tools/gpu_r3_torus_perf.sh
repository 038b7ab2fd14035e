#!/bin/bash
# round 3: torus timings (plain two-limb kernels, unrolled kernel) + phase profile of the unrolled torus kernel
mkdir -p gpurun_out
L=gpurun_out/r3_torus_perf.log
: > $L
timeout -k 10 200 python tools/br_timing.py 1,256,8192 0 65 2>&1 | grep -v amdgpu.ids >> $L &&
BMI_UNROLL=2 timeout -k 10 200 python tools/br_timing.py 1,64,128,256,512,8192 0 65 2>&1 | grep -v amdgpu.ids >> $L &&
make -C bounty-matrix-inversion_amd/csrc prof -j8 > gpurun_out/r3_make_prof.log 2>&1 &&
timeout -k 10 200 python tools/phase_prof.py 1 2 10 2 65 2>&1 | grep -v amdgpu.ids >> $L &&
timeout -k 10 200 python tools/phase_prof.py 256 2 10 2 65 2>&1 | grep -v amdgpu.ids >> $L
rc=$?
cat $L
exit $rc
