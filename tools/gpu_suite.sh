#!/bin/bash
# the whole GPU suite as the driver runs it (cold program cache), with durations; output goes straight to a file under
# gpurun_out/ (unbuffered: a silent run is taken to be hung after 7 minutes)
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 ${BMI_T:-1100} python -u -m pytest tests/ -x -q -m gpu --durations=25 ${BMI_PYTEST_ARGS} > gpurun_out/gpu_suite.log 2>&1
rc=$?
grep -v amdgpu.ids gpurun_out/gpu_suite.log | tail -45
exit $rc
