#!/bin/bash
# the floating-point-transform torus kernels: their own test file, then the default bench line (torus headline)
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 900 python -u -m pytest tests/test_gpu_torus_fft.py tests/test_gpu_c_abi.py -m gpu -x -q -s > gpurun_out/r3_fft_tests.log 2>&1 || { grep -v amdgpu.ids gpurun_out/r3_fft_tests.log | tail -40; exit 1; }
grep -v amdgpu.ids gpurun_out/r3_fft_tests.log | grep "largest\|passed\|failed" | tail -8
T0=$(date +%s)
timeout -k 10 900 python bench.py --inverse-sizes ${BMI_INV_SIZES:-2,3} > gpurun_out/r3_fft_bench.json 2> gpurun_out/r3_fft_bench.err || { tail -5 gpurun_out/r3_fft_bench.err; exit 1; }
echo "bench took $(( $(date +%s) - T0 )) s"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3_fft_bench.json"))
keep = {k: d.get(k) for k in ("value", "ms_per_step", "latency_ms_1", "latency_ms_256", "value_torus64", "frac_torus64", "alu_frac_torus64", "value_p49", "frac_p49", "alu_frac_p49", "latency_ms_p49", "value_torus64_unrolled", "latency_ms_torus64_unrolled", "inverse_3x3_s_torus64")}
print(json.dumps(keep))
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "kernel", "kernel_ms", "traffic")}, "alu", {k: d["roofline"]["alu"].get(k) for k in ("frac", "sclk_mhz", "error")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("gpu_matches_bit_for_bit"), "noise", d["config"]["output_noise"])
for k in d["config"]:
    if k.startswith("encrypted_inverse"):
        v = d["config"][k]
        print(k, {s: (x.get("evaluate_s"), x.get("depth")) for s, x in v.items()} if isinstance(v, dict) and "error" not in v else v)
PY
