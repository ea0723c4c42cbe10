#!/bin/bash
# PMC passes over a few training steps (separate --pmc runs), summarised per kernel by tools/pmc_summary.py.
export TMPDIR=/tmp
OUT=${1:-gpurun_out/train_pmc}
mkdir -p $OUT
for C in "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "MfmaUtil" "TA_BUSY_avr TCC_BUSY_avr TCC_TAG_STALL_sum" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 tools/train_bench.py --steps 6 --warmup 2 > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err || echo "pass $N failed"
done
python tools/pmc_summary.py $OUT $OUT/pmc_summary.json
rm -rf $OUT/pmc_*/
