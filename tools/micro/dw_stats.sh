# per-kernel averages of the eager 1024-ray training step (rocprofv3 --stats), for the library NERF_AMD_LIB names
export TMPDIR=/tmp
for prec in bf16 fp32_split; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dwstat_$prec -- python3 tools/train_bench.py --steps 30 --precision $prec > gpurun_out/dwstat_$prec.json 2>/dev/null
grep -h "dw_multi\|mlp_bwd\|mlp_bf16_s16\|mlp_split" gpurun_out/dwstat_$prec/*/*_kernel_stats.csv | sed 's/"\([^"(<]*\)[^"]*"/\1/' | cut -d, -f1-7
rm -rf gpurun_out/dwstat_$prec
done
