#!/bin/bash
# round 3 verdict 1(b): the encoder-layer tail as 32-row workgroups (<= 80 KB LDS, 128 VGPR: two per CU) against the 64-row form
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4tail32; mkdir -p $O
VOG_TAIL_ROWS32=2 python -m pytest tests/test_gpu_ops.py -x -q -k "tail" 2>&1 | tail -3 > $O/tests.log
VOG_TAIL_ROWS32=2 python -m pytest tests/test_gpu_forward.py -x -q -k "full_vs_reference or small_vs_reference" 2>&1 | tail -3 >> $O/tests.log
run() { timeout 300 python bench.py --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg2 256 CUs: 64-row tails            -> $(run --steps 800 --warmup 80)"
echo "cfg2 256 CUs: mul tail 32 rows        -> $(VOG_TAIL_ROWS32=1 run --steps 800 --warmup 80)"
echo "cfg2 128 CUs: 64-row tails            -> $(HSA_CU_MASK=0:0-127 run --steps 800 --warmup 80)"
echo "cfg2 128 CUs: mul tail 32 rows        -> $(HSA_CU_MASK=0:0-127 VOG_TAIL_ROWS32=1 run --steps 800 --warmup 80)"
echo "cfg4: 64-row tails                    -> $(run --workload cfg4 --steps 400 --warmup 40)"
echo "cfg4: mul tail 32 rows                -> $(VOG_TAIL_ROWS32=1 run --workload cfg4 --steps 400 --warmup 40)"
echo "cfg4: both tails 32 rows (unpaired obj tail) -> $(VOG_TAIL_ROWS32=2 run --workload cfg4 --steps 400 --warmup 40 --set pair_mask=5)"
done
echo "cfg2 streams=1: 64 rows $(run --streams 1 --steps 400 --warmup 40)  32 rows $(VOG_TAIL_ROWS32=1 run --streams 1 --steps 400 --warmup 40)"
} > $O/tail32.log 2>&1
bash scratch/kt_forward.sh cfg4 6 0 2>/dev/null | grep tx_tail > $O/kt64.txt
VOG_TAIL_ROWS32=1 bash scratch/kt_forward.sh cfg4 6 0 2>/dev/null | grep tx_tail > $O/kt32.txt
cat $O/tests.log $O/tail32.log; echo "cfg4 kernel times 64 rows:"; cat $O/kt64.txt; echo "32 rows:"; cat $O/kt32.txt
