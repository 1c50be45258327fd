#!/bin/bash
# experiment 3b: hipGraphLaunch ignores a stream's CU mask (3: identical numbers down to 1/4 of the CUs), so the same question
# with eager launches (--no-graph): every kernel goes to the masked stream itself
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4cumask; mkdir -p $O
run() { timeout 120 python bench.py --steps 800 --warmup 80 --throughput-only --no-graph "$@" 2>&1 | tail -1; }
{
for s in 1 2 3 4; do echo "eager streams=$s plain streams -> $(run --streams $s)"; done
for m in ffffffff 77777777 55555555; do
echo "eager streams=1 mask=$m -> $(VOG_BENCH_CU_MASK=$m run --streams 1)"
echo "eager streams=2 mask=$m -> $(VOG_BENCH_CU_MASK=$m run --streams 2)"
done
for m in ffffffff 77777777; do
echo "eager streams=3 mask=$m -> $(VOG_BENCH_CU_MASK=$m run --streams 3)"
done
echo "eager streams=4 mask=ffffffff -> $(VOG_BENCH_CU_MASK=ffffffff run --streams 4)"
echo "eager streams=2 disjoint halves -> $(VOG_BENCH_CU_MASK=55555555,aaaaaaaa run --streams 2)"
} > $O/cumask_eager.log 2>&1
cat $O/cumask_eager.log
