# the 1024-ray training step for the stonehenge encodings (multires 15 / 6) beside the default 10 / 4, both precisions
export TMPDIR=/tmp
for prec in bf16 fp32_split; do
for mr in "10 4" "15 6"; do
set -- $mr
echo "== $prec multires $1 / $2"
python3 tools/train_bench.py --steps 100 --precision $prec --multires $1 --multires-views $2 --graph | cut -c1-110
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t156 -- python3 tools/train_bench.py --steps 30 --precision $prec --multires $1 --multires-views $2 > /dev/null 2>&1
grep -h "dw_multi\|mlp_bwd\|mlp_bf16_s16\|mlp_split" gpurun_out/t156/*/*_kernel_stats.csv | sed 's/"\([^"(<]*\)[^"]*"/\1/' | cut -d, -f1-4
rm -rf gpurun_out/t156
done
done
