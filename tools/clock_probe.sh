# samples GPU clocks / power while bench.py runs (read-only queries)
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -u bench.py --no-cpu-baseline --steps 12 > gpurun_out/clock_bench.log 2>&1 &
BP=$!
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | head -8
  echo ---
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/clock_bench.log | cut -c1-120
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
