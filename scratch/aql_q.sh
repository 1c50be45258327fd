#!/bin/bash
for q in 3 4 6 8 12 16; do
  r=$(timeout 300 python bench.py --steps 480 --warmup 48 --mode aql --queues $q --interleave 1 --throughput-only 2>&1 | tail -1)
  echo "queues=$q interleave=1 -> $r"
done
