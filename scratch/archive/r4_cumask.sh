#!/bin/bash
# round 4, experiment 3: is the 4-stream loop bound by CU time? the same loop with every stream restricted to a fraction of
# the CUs (a repeating bit pattern: the same fraction on every XCD / shader engine whatever the bit order)
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4cumask; mkdir -p $O
run() { python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>&1 | tail -1; }
{
for s in 4 3 1; do
echo "streams=$s all CUs (plain streams)   -> $(run --streams $s)"
for m in ffffffff 77777777 55555555 11111111; do
echo "streams=$s mask=$m                -> $(VOG_BENCH_CU_MASK=$m run --streams $s)"
done
done
echo "streams=4 disjoint quarters         -> $(VOG_BENCH_CU_MASK=11111111,22222222,44444444,88888888 run --streams 4)"
echo "streams=4 disjoint halves (2+2)     -> $(VOG_BENCH_CU_MASK=55555555,aaaaaaaa,55555555,aaaaaaaa run --streams 4)"
} > $O/cumask.log 2>&1
cat $O/cumask.log
