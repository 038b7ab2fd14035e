# usage (on the GPU box): bash tools/gpu_ab.sh "name1 name2 ..." [batches] [variant] [q_bits] [rounds]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for round in $(seq 1 ${5:-2}); do
  for n in $1; do
    echo "== $n"
    BMI_TFHE_LIB=$GRAFT_REPO_ROOT/bounty-matrix-inversion_amd/lib/ab_$n.so timeout -k 10 200 python tools/br_timing.py ${2:-8192} ${3:-0} ${4:-49} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab.log
  done
done
