#!/bin/bash
# Builds A/B variants of the kernel library: tools/ab_build.sh name1:"flags" name2:"flags" ...  -> lib/ab_<name>.so
# (flags replace nothing: they are appended to the normal compile line; use SCHED=<strategy> inside flags to swap the scheduler)
set -e
cd "$(dirname "$0")/../bounty-matrix-inversion_amd/csrc"
for cfg in "$@"; do
  n=${cfg%%:*}; f=${cfg#*:}
  sched="-mllvm -amdgpu-sched-strategy=max-ilp"
  case "$f" in *SCHED=default*) sched=""; f=${f/SCHED=default/};; *SCHED=*) s=$(echo "$f" | sed 's/.*SCHED=\([a-z-]*\).*/\1/'); sched="-mllvm -amdgpu-sched-strategy=$s"; f=$(echo "$f" | sed 's/SCHED=[a-z-]*//');; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $sched $f -ffp-contract=off -shared -no-hip-rt -o ../lib/ab_$n.so -x hip bmi_kernels.hip -x hip bmi_kernels_f64.hip -x hip bmi_kernels_f64u.hip -x hip bmi_kernels_t64.hip -x hip bmi_host.cpp -x hip bmi_compile.cpp -lpthread
  echo "built ab_$n.so ($sched $f)"
done
