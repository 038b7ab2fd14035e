set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_inverse.py -m gpu -x -q -k "readme_low" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 600 python bench.py --no-cpu-baseline --no-second-field --inverse-sizes 2 --steps 2 > gpurun_out/bench_readme.json 2> gpurun_out/bench_readme.err || true
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_readme.json'))
print(json.dumps(d['config'].get('reference_readme_benchmark'), indent=1)[:3000])
PY
