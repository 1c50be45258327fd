#!/bin/bash
for f in 1 0 2 11 21; do for q in 1 4; do for k in 1 4; do
  r=$(VOG_PERF_EXPERIMENTS=1 VOG_AQL_FENCE=$f timeout 300 python bench.py --steps 480 --warmup 48 --mode aql --queues $q --interleave $k --throughput-only 2>&1 | tail -1)
  echo "fence=$f queues=$q interleave=$k -> $r"
done; done; done
