#!/bin/bash
# Sample GPU clock / power while the training step runs (is the step power-bound like the field kernel?).
# usage: tools/micro/power_trace_train.sh > gpurun_out/power_trace_train.txt
python tools/train_bench.py --steps 12000 --rays ${1:-1024} > /tmp/train_power.json 2>/dev/null &
B=$!
sleep 8
for i in $(seq 1 12); do
  if ! kill -0 $B 2>/dev/null; then break; fi
  echo "--- sample $i"
  rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use|busy" | head -8
  sleep 0.4
done
wait $B
cut -c1-120 /tmp/train_power.json
