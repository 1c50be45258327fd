#!/bin/bash
# round 4: E x F separable attention at p100 (cfg 4); skinny register variants (pair keeps 4 QKV workgroups per CU)
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4ef; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "struct or skinny or gemm" 2>&1 | tail -5 > $O/tests_ops.log
python -m pytest tests/test_gpu_forward.py -x -q -k "p100 or golden" 2>&1 | tail -5 > $O/tests_fwd.log
run() { timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg4: EF attention (default)  -> $(run --workload cfg4)"
echo "cfg4: old struct_lds          -> $(VOG_ATTN_STRUCT_EF=0 run --workload cfg4)"
echo "cfg2: default                 -> $(run --steps 800 --warmup 80)"
echo "cfg2: pair_mask=3             -> $(run --steps 800 --warmup 80 --set pair_mask=3)"
echo "cfg2 128 CUs: default         -> $(HSA_CU_MASK=0:0-127 run --steps 800 --warmup 80)"
echo "cfg2 128 CUs: pair_mask=3     -> $(HSA_CU_MASK=0:0-127 run --steps 800 --warmup 80 --set pair_mask=3)"
done
echo "cfg4 streams=1: EF $(run --workload cfg4 --streams 1) old $(VOG_ATTN_STRUCT_EF=0 run --workload cfg4 --streams 1)"
echo "cfg3: $(run --workload cfg3)   cfg5: $(run --workload cfg5)"
} > $O/ef.log 2>&1
cat $O/tests_ops.log $O/tests_fwd.log $O/ef.log
