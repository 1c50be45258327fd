#!/bin/bash
# cfg 4: per-kernel durations + counters of the E x F attention kernel
R=$PWD; O=$R/gpurun_out/r4efprof; mkdir -p $O; export TMPDIR=/tmp
bash scratch/kt_forward.sh cfg4 6 0 > $O/kt_cfg4.txt 2>&1
cd /tmp
i=0
for c in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -o r -- python $R/scratch/prof_forward.py cfg4 3 0 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scratch/pmc_kernels.py $f attn_struct_ef "tx_tail_kernel<BF16, F16, 3" attn_tile2 attn_struct_lds >> $O/pmc.txt 2>&1
done
find $O -name "*.csv" -size +5M -delete
cat $O/kt_cfg4.txt $O/pmc.txt
