#!/bin/bash
for g in 1 2 4; do for s in 1 2 3 4; do
  r=$(timeout 300 python bench.py --steps 480 --warmup 48 --cobatch $g --streams $s --throughput-only 2>&1 | tail -1)
  echo "graph cobatch=$g streams=$s -> $r"
done; done
for g in 4; do for q in 1 2 4; do
  r=$(timeout 300 python bench.py --steps 480 --warmup 48 --mode aql --cobatch $g --queues $q --throughput-only 2>&1 | tail -1)
  echo "aql cobatch=$g queues=$q -> $r"
done; done
