#!/bin/bash
# throughput vs HIP streams (batches in flight): gpurun -- bash scratch/streams_sweep.sh
for s in 1 2 3 4 5 6 8; do
  r=$(python bench.py --throughput-only --streams $s 2>/dev/null | tail -1)
  echo "streams=$s -> $r"
done
