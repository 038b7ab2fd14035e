#!/bin/bash
mkdir -p gpurun_out
make -C bounty-matrix-inversion_amd/csrc prof -j8 > gpurun_out/r3_make_prof.log 2>&1 || { tail gpurun_out/r3_make_prof.log; exit 1; }
timeout -k 5 120 python tools/phase_prof.py 1 2 10 2 49 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_pipe_prof.log
cat gpurun_out/r3_pipe_prof.log
