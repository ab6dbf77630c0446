# sample the GPU clocks / power while the bench's timed loop is running (is the box power-throttled?)
( python bench.py --steps 600 --warmup 3 --no-cpu-baseline > /tmp/clk_bench.json 2>/dev/null ) &
BP=$!
sleep 14
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.4
done
wait $BP
tail -1 /tmp/clk_bench.json | cut -c1-200
echo "idle:"; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
