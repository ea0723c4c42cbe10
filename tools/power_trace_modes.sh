#!/bin/bash
# Socket power and shader clock while each field kernel runs, in ONE session on one device: the bf16 kernel, the
# split-precision kernel and the exact-fp32 kernel (bench.py --precision ..., sub-records off).
# usage: tools/power_trace_modes.sh > gpurun_out/power_trace_modes.txt
for PREC in bf16 fp32_split fp32; do
  case $PREC in bf16) STEPS=300;; fp32_split) STEPS=90;; fp32) STEPS=24;; esac
  python bench.py --precision $PREC --steps $STEPS --warmup 2 --no-cpu-baseline --no-subrecords > /tmp/bench_power_$PREC.json 2>/dev/null &
  BENCH=$!
  sleep 6
  echo "=== $PREC"
  for i in $(seq 1 8); do
    if ! kill -0 $BENCH 2>/dev/null; then break; fi
    rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ' '; echo
    sleep 0.5
  done
  wait $BENCH
  python -c "import sys,json; d=json.loads(open('/tmp/bench_power_$PREC.json').read().strip().splitlines()[-1]); print('bench', '$PREC', round(d['value']), 'rays/s', round(d['roofline']['achieved'],1), 'TFLOP/s', round(d['roofline']['frac'],3), 'of', d['roofline']['peak'])"
  sleep 3
done
echo "=== idle"
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ' '; echo
