#!/bin/bash
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4ef64; mkdir -p $O
VOG_ATTN_STRUCT_EF=4 python -m pytest tests/test_gpu_ops.py -x -q -k "struct" 2>&1 | tail -4 > $O/tests_ops.log
VOG_ATTN_STRUCT_EF=4 python -m pytest tests/test_gpu_forward.py -x -q -k "p100" 2>&1 | tail -4 > $O/tests_fwd.log
( cd scratch; ./ts_attn_ef | head -1; VOG_ATTN_STRUCT_EF=4 ./ts_attn_ef | head -1 ) > $O/ts.log 2>&1
run() { timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only --workload cfg4 "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg4: EF 32-proposal blocks -> $(run)"
echo "cfg4: EF 64-proposal blocks -> $(VOG_ATTN_STRUCT_EF=4 run)"
done
} > $O/ef.log 2>&1
cat $O/tests_ops.log $O/tests_fwd.log $O/ts.log $O/ef.log
