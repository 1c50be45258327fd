#!/bin/bash
for dag in "" "--dag"; do for s in 1 2 3 4 5 6 8; do
  r=$(python bench.py --steps 400 --warmup 40 --streams $s $dag --throughput-only 2>/dev/null | tail -1)
  echo "dag=[$dag] streams=$s -> $r"
done; done
