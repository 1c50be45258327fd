#!/bin/bash
# round 4: lean (<= 128 VGPR, 4 workgroups / CU) separable attention, pair_mask, on the whole chip and on half of it
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4lean1; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -k "struct or golden or rel_attention" 2>&1 | tail -4 > $O/tests.log
run() { timeout 200 python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "256 CUs: lean struct1 (default)     -> $(run)"
echo "256 CUs: old struct1                -> $(VOG_ATTN_STRUCT1_LEAN=0 run)"
echo "256 CUs: lean + pair_mask=3         -> $(run --set pair_mask=3)"
echo "256 CUs: lean + qkv_lean            -> $(run --set qkv_lean=1)"
echo "128 CUs: lean struct1 (default)     -> $(HSA_CU_MASK=0:0-127 run)"
echo "128 CUs: old struct1                -> $(HSA_CU_MASK=0:0-127 VOG_ATTN_STRUCT1_LEAN=0 run)"
echo "128 CUs: lean + pair_mask=3         -> $(HSA_CU_MASK=0:0-127 run --set pair_mask=3)"
echo "128 CUs: lean + qkv_lean            -> $(HSA_CU_MASK=0:0-127 run --set qkv_lean=1)"
done
echo "streams=1: lean $(run --streams 1) old $(VOG_ATTN_STRUCT1_LEAN=0 run --streams 1)"
} > $O/lean1.log 2>&1
cat $O/tests.log $O/lean1.log
