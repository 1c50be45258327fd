#!/bin/bash
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4c4b; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "struct or tail" 2>&1 | tail -3 > $O/tests.log
python -m pytest tests/test_gpu_forward.py -x -q -k "p100" 2>&1 | tail -3 >> $O/tests.log
run() { timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only --workload cfg4 "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg4 default (EF Q via LDS, tail nt)  -> $(run)"
echo "cfg4 tail nt off                      -> $(VOG_TAIL_NT=0 run)"
echo "cfg4 qkv_lean=1                       -> $(run --set qkv_lean=1)"
done
} > $O/c4.log 2>&1
bash scratch/kt_forward.sh cfg4 6 0 > $O/kt_cfg4.txt 2>&1
VOG_TAIL_NT=0 bash scratch/kt_forward.sh cfg4 6 0 > $O/kt_cfg4_nt0.txt 2>&1
cat $O/tests.log $O/c4.log $O/kt_cfg4.txt; grep tx_tail $O/kt_cfg4_nt0.txt
