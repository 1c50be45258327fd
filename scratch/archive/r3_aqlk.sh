#!/bin/bash
for m in graph aql; do for k in "20 5" "40 5" "400 40"; do set -- $k
  echo -n "$m K=$1: "; python3 bench.py --gpus 1 --steps $1 --warmup $2 --mode $m --throughput-only 2>/dev/null | tail -1
done; done
