set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${BMI_T:-800} python -m pytest tests/test_gpu_inverse.py -m gpu -x -q -k "torus64" --durations=6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gpu_t64inv.log | tail -14
timeout -k 10 300 python bench.py --q-bits 65 --no-cpu-baseline --inverse-sizes 2,3,4 --steps 3 > gpurun_out/bench_t64inv.json 2> gpurun_out/bench_t64inv.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_t64inv.json'))
print(d['value'], d['roofline']['frac'])
for k,v in d['config']['encrypted_inverse_wall_clock'].items(): print(k, v['evaluate_s'], v['end_to_end_s'], v['ms_per_level'], v['matches_plaintext_circuit'])
PY
