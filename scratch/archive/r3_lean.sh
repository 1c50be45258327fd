#!/bin/bash
for rep in 1 2 3; do for wl in cfg2 cfg3 cfg5; do for ql in 0 1; do
  echo -n "$wl qkv_lean=$ql: "; python3 bench.py --gpus 1 --workload $wl --steps 400 --warmup 40 --throughput-only --set qkv_lean=$ql 2>/dev/null | tail -1
done; done; done
