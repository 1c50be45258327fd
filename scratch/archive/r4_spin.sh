#!/bin/bash
A="--gpus 1 --steps 20 --warmup 5 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2 3 4; do echo -n "default:            "; python bench.py $A 2>/dev/null | tail -1; done
for r in 1 2 3 4; do echo -n "HSA_ENABLE_INTERRUPT=0: "; HSA_ENABLE_INTERRUPT=0 python bench.py $A 2>/dev/null | tail -1; done
A4="--gpus 1 --steps 400 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do echo -n "K=400 default:            "; python bench.py $A4 2>/dev/null | tail -1; done
for r in 1 2; do echo -n "K=400 HSA_ENABLE_INTERRUPT=0: "; HSA_ENABLE_INTERRUPT=0 python bench.py $A4 2>/dev/null | tail -1; done
