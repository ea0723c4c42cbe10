#!/bin/bash
# A/B of scratch builds of the training forward (scratch_libs/*.so, made with make OUT=... EXTRA=...): step time without the
# profiler, then the kernels' average durations under rocprofv3 --stats.
export TMPDIR=/tmp
for L in default "$@"; do
  if [ "$L" = default ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$GRAFT_REPO_ROOT/scratch_libs/$L; fi
  echo "== $L"
  python3 tools/train_bench.py --steps 200 | cut -c1-70
  rm -rf /tmp/sv && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sv -- python3 tools/train_bench.py --steps 40 > /dev/null 2>&1
  grep -h "mlp_bf16_s16_kernel\|mlp_bwd_s16_kernel" /tmp/sv/*/*_kernel_stats.csv | sed -E 's/^"(.{0,60})[^"]*",([0-9]+),([0-9]+),([0-9.]+),.*/\1 calls \2 avg_ns \4/'
done
