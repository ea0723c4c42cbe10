# every kernel of the eager 1024-ray training step (rocprofv3 --stats): train_stats.sh <precision> <multires> <multires_views>
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tstat -- python3 tools/train_bench.py --steps 30 --precision $1 --multires $2 --multires-views $3 > /dev/null 2>&1
sed 's/"\([^"(<]*\)[^"]*"/\1/' gpurun_out/tstat/*/*_kernel_stats.csv | cut -d, -f1-5 | head -24
rm -rf gpurun_out/tstat
