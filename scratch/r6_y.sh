#!/bin/bash
# which HIP streams overlap: the 4-stream loop on different members of the pool
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
echo "streams 3: $($B --streams 3 2>/dev/null)"
export VOG_PERF_EXPERIMENTS=1
for ids in 0,1 0,2 0,3 0,4 1,2 1,3 2,3 0,5 ; do echo "2 streams $ids: $(VOG_BENCH_STREAM_IDS=$ids $B --streams 2 2>/dev/null)"; done
for ids in 0,1,2,3 1,2,3,4 0,2,3,5 4,5,6,7 0,1,2,3,4,5,6,7; do echo "4 streams $ids: $(VOG_BENCH_STREAM_IDS=$ids $B --streams 4 2>/dev/null)"; done
