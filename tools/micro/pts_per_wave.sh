#!/bin/bash
# Build and run tools/micro/pts_per_wave.hip with a power / clock sampler beside it: the sampler's lines land between the
# BEGIN / END lines of the arrangement that was running.   usage: tools/micro/pts_per_wave.sh [seconds per arrangement]
set -o pipefail
SEC=${1:-2.5}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/ppw tools/micro/pts_per_wave.hip || exit 1
/tmp/ppw $SEC &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed -E 's/^GPU\[0\]\s*: //' | tr '\n' ' ' | sed 's/$/\n/'
  sleep 0.35
done
wait $PID
echo "--- idle"; sleep 1.5
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | sed -E 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo
