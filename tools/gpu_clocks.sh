# samples GPU clock / power while the throughput kernel runs (is the f64 kernel power-limited?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 200 python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-inverse --no-second-field > gpurun_out/clk_bench.log 2>&1) &
BP=$!
sleep 30
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|Power|Temperature|junction" | head -8
  echo ---
  sleep 1
done
wait $BP
tail -c 300 gpurun_out/clk_bench.log
