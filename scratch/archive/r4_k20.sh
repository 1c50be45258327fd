#!/bin/bash
A="--gpus 1 --steps 20 --warmup 5 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2 3 4 5; do echo -n "bind: "; python bench.py $A 2>/dev/null | tail -1; done
for r in 1 2 3 4 5; do echo -n "nobind: "; VOG_BENCH_NUMA_BIND=0 python bench.py $A 2>/dev/null | tail -1; done
