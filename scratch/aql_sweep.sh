#!/bin/bash
for split in "" "--no-split"; do for q in 1 2 4; do for k in 1 2 3 4; do
  r=$(timeout 300 python bench.py --steps 480 --warmup 48 --mode aql --queues $q --interleave $k $split --throughput-only 2>&1 | tail -1)
  echo "split=[$split] queues=$q interleave=$k -> $r"
done; done; done
