python bench.py --no-cpu-baseline --steps 150 --warmup 3 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
: > gpurun_out/clk_samples.txt
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+)/power \1/' | tr '\n' ' ' >> gpurun_out/clk_samples.txt
  echo >> gpurun_out/clk_samples.txt
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' gpurun_out/clk_bench.log
sort -t' ' -k4 -n gpurun_out/clk_samples.txt | tail -12
wc -l gpurun_out/clk_samples.txt
