#!/bin/bash
# A/B of the torus floating-point-transform kernel: timing of experiment builds (some return wrong words on purpose), then counters
mkdir -p gpurun_out
export TMPDIR=/tmp
LOG=gpurun_out/r3_t64f_ab.log
: > $LOG
for lib in ${BMI_AB_LIBS:-libbmi_tfhe.so}; do
  echo "== $lib" | tee -a $LOG
  BMI_TFHE_LIB=$PWD/bounty-matrix-inversion_amd/lib/$lib timeout -k 5 200 python tools/br_timing.py 8192 5 65 2>&1 | grep --line-buffered -v amdgpu.ids | cut -c1-200 | tee -a $LOG
done
if [ -n "$BMI_AB_PMC" ]; then
  rm -rf gpurun_out/t64f_pmc
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/t64f_pmc/a -- python3 tools/br_timing.py 8192 5 65 > gpurun_out/t64f_pmc_a.log 2>&1 || echo "pmc a failed"
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/t64f_pmc/b -- python3 tools/br_timing.py 8192 5 65 > gpurun_out/t64f_pmc_b.log 2>&1 || echo "pmc b failed"
  python3 - <<'PY' | tee -a gpurun_out/r3_t64f_ab.log
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/t64f_pmc/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "blind_rotate" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in acc.items():
            print(d, k, {c: f"{x:.4g}" for c, x in v.items()})
PY
  rm -rf gpurun_out/t64f_pmc
fi
