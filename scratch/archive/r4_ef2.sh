#!/bin/bash
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4ef2; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "struct" 2>&1 | tail -3 > $O/tests_ops.log
python -m pytest tests/test_gpu_forward.py -x -q -k "p100" 2>&1 | tail -3 > $O/tests_fwd.log
run() { timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg4: EF v2, 1 wg/CU  -> $(run --workload cfg4)"
echo "cfg4: EF v2, 2 wg/CU  -> $(VOG_ATTN_STRUCT_EF=2 run --workload cfg4)"
echo "cfg4: old struct_lds  -> $(VOG_ATTN_STRUCT_EF=0 run --workload cfg4)"
done
echo "cfg4 streams=1: EF1 $(run --workload cfg4 --streams 1) EF2 $(VOG_ATTN_STRUCT_EF=2 run --workload cfg4 --streams 1) old $(VOG_ATTN_STRUCT_EF=0 run --workload cfg4 --streams 1)"
} > $O/ef.log 2>&1
bash scratch/kt_forward.sh cfg4 6 0 > $O/kt_cfg4.txt 2>&1
cat $O/tests_ops.log $O/tests_fwd.log $O/ef.log; grep -E "struct_ef|sum" $O/kt_cfg4.txt
