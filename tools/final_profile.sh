#!/bin/bash
# Round-end measurement pass on the GPU box: the bench line and the rocprofv3 summaries that go under profiles/.
#   tools/final_profile.sh <tag>      -> gpurun_out/final_<tag>/...
#   TRAIN_ONLY=1 tools/final_profile.sh <tag>     only the training-step part (when only the backward's sources changed)
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$TRAIN_ONLY" ]; then
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench.json
# kernel trace + stats of the same command (no CPU leg: it only adds host time; no sub-records, so the field kernel's
# average in the stats is over the metric's own launches -- 10 per view, warm-up included -- and must agree with the
# bench's hipEvent figure, roofline.avg_launch_ms)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --no-subrecords > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
# counters: separate passes, --pmc only
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "MfmaUtil" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || echo "pmc pass $N failed"
done
fi
# the training step: the timed lines -- eager and captured (utils.CapturedTrainStep), bf16 and split precision --
# then the kernel stats and the timeline of two steps under the profiler
python tools/train_bench.py --steps 300 > $OUT/train_bench.json 2> $OUT/train_bench.err || echo "train bench failed"
python tools/train_bench.py --steps 300 --graph > $OUT/train_bench_graph.json 2> $OUT/train_bench_graph.err || echo "train bench (graph) failed"
python tools/train_bench.py --steps 200 --precision fp32_split > $OUT/train_bench_split.json 2> $OUT/train_bench_split.err || echo "train bench (split) failed"
python tools/train_bench.py --steps 200 --precision fp32_split --graph > $OUT/train_bench_split_graph.json 2> $OUT/train_bench_split_graph.err || echo "train bench (split, graph) failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_split -- python3 tools/train_bench.py --steps 50 --precision fp32_split > $OUT/train_split_under_rocprof.json 2> $OUT/train_split.err || echo "split train profile failed"
cp $OUT/train_split/*/*_kernel_stats.csv $OUT/train_split_kernel_stats.csv
rm -rf $OUT/train_split
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 tools/train_bench.py --steps 50 > $OUT/train_under_rocprof.json 2> $OUT/train.err || echo "train profile failed"
cp $OUT/train/*/*_kernel_stats.csv $OUT/train_kernel_stats.csv
python tools/step_timeline.py $(find $OUT/train -name "*kernel_trace.csv" | head -1) > $OUT/train_timeline.txt
rm -rf $OUT/train
# keep what is judged (the summaries) and drop the per-dispatch tables: gpurun merges at most 64 MiB back
if [ -z "$TRAIN_ONLY" ]; then
python tools/pmc_summary.py $OUT $OUT/pmc_summary.json
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_*/
fi
echo final-profile-done
