#!/bin/bash
# PMC passes over the fused kernels (scratch/mb_tail.py): gpurun -- bash scratch/prof_tail.sh
R=$PWD; O=$R/gpurun_out/pmc_tail; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -o r -- python $R/scratch/mb_tail.py "$@" > $O/sq.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY TCC_HIT TCC_MISS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/tc -o r -- python $R/scratch/mb_tail.py "$@" > $O/tc.log 2>&1
cd $R
for d in sq tc; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python scratch/pmc_kernels.py $f tail vis_enc attn lstm_layer gemm > $O/$d.md; cat $O/$d.md; done
tail -3 $O/sq.log $O/tc.log
