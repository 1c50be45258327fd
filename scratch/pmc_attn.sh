#!/bin/bash
# counters of the long-sequence attention kernels: gpurun -- bash scratch/pmc_attn.sh
R=$PWD; O=$R/gpurun_out/pmc_attn; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python $R/scratch/mb_attn_long.py 20 both
i=0
for c in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -o r -- python $R/scratch/mb_attn_long.py 3 both > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scratch/pmc_kernels.py $f attn_tile
done
