#!/bin/bash
# Sample GPU clock / power while bench.py runs (evidence for the "power-bound" reading of the MFMA kernel).
# usage: tools/power_trace.sh > gpurun_out/power_trace.txt
python bench.py --steps 500 --no-cpu-baseline > /tmp/bench_power.json 2>/dev/null &
BENCH=$!
sleep 7
for i in $(seq 1 16); do
  if ! kill -0 $BENCH 2>/dev/null; then break; fi
  echo "--- sample $i"
  rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use|busy" | head -8
  sleep 0.4
done
wait $BENCH
echo "--- idle"
sleep 2
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4
cat /tmp/bench_power.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['achieved'], d['roofline']['frac'])"
