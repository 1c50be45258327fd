#!/bin/bash
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4c4; mkdir -p $O
run() { VOG_ATTN_STRUCT_EF=2 timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only --workload cfg4 "$@" 2>/dev/null | tail -1; }
{
echo "cfg4 EF(2/CU) default       -> $(run)"
echo "cfg4 EF(2/CU) qkv_lean=1    -> $(run --set qkv_lean=1)"
echo "cfg4 EF(2/CU) pair_mask=3   -> $(run --set pair_mask=3)"
echo "cfg4 EF(2/CU) pair_launches=0 -> $(run --set pair_launches=0)"
echo "cfg4 EF(2/CU) streams=3     -> $(run --streams 3)"
echo "cfg4 EF(2/CU) streams=2     -> $(run --streams 2)"
echo "cfg4 EF(2/CU) default       -> $(run)"
} > $O/c4.log 2>&1
cat $O/c4.log
