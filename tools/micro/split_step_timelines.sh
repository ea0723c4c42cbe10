# kernel timelines of the split-precision training step, eager and captured (why is the captured one not faster?)
export TMPDIR=/tmp
for mode in eager graph; do
  flag=""; [ $mode = graph ] && flag="--graph"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sst_$mode -- python3 tools/train_bench.py --steps 30 --precision fp32_split $flag > gpurun_out/sst_$mode.json 2>/dev/null
  python tools/step_timeline.py $(find gpurun_out/sst_$mode -name "*kernel_trace.csv" | head -1) > gpurun_out/sst_$mode.txt
  rm -rf gpurun_out/sst_$mode
done
