#!/bin/bash
A="--gpus 1 --steps 20 --warmup 5 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for b in 0 20 100 500; do for r in 1 2 3; do echo -n "burn-in $b ms: "; VOG_BENCH_BURNIN_MS=$b python bench.py $A 2>/dev/null | tail -1; done; done
