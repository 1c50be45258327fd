#!/bin/bash
# experiment 3c: process-wide CU masks (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) - the per-stream masks of 3 / 3b were not applied
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4cumask; mkdir -p $O
run() { timeout 120 python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>&1 | tail -1; }
{
for s in 1 2 3 4; do
echo "streams=$s all CUs -> $(run --streams $s)"
echo "streams=$s HSA_CU_MASK=0:0-191 -> $(HSA_CU_MASK=0:0-191 run --streams $s)"
echo "streams=$s HSA_CU_MASK=0:0-127 -> $(HSA_CU_MASK=0:0-127 run --streams $s)"
done
echo "streams=1 ROC_GLOBAL_CU_MASK 128 bits -> $(ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff run --streams 1)"
echo "streams=2 ROC_GLOBAL_CU_MASK 128 bits -> $(ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff run --streams 2)"
} > $O/cumask_global.log 2>&1
cat $O/cumask_global.log
