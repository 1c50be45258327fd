#!/bin/bash
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do
for v in 0 1; do for s in 2 3 4; do
echo -n "pair_attn $v cfg4 $s streams: "; python bench.py $A --workload cfg4 --steps 200 --warmup 20 --streams $s --set pair_attn=$v 2>/dev/null | tail -1
done; done; done
