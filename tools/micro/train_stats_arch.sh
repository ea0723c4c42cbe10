# every kernel of the eager 1024-ray training step for another architecture: train_stats_arch.sh "<train_bench flags>"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tstat -- python3 tools/train_bench.py --steps 10 $1 > /dev/null 2>&1
sed 's/"\([^"(<]*\)[^"]*"/\1/' gpurun_out/tstat/*/*_kernel_stats.csv | cut -d, -f1-5 | head -14
rm -rf gpurun_out/tstat
