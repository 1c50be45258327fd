#!/bin/bash
# round 5: upper bound of "layer-1 input projection split over helper workgroups": run HALF of it (wrong results)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
for r in 1 2 3; do
for h in 0 1; do
echo -n "half_proj $h kernels: "; VOG_LSTM_HALF_PROJ=$h python scratch/mb_tail.py "lstm_layer+obj_tail" "lstm_layer#1" 2>/dev/null | tail -1
echo -n "half_proj $h cfg2 4 streams: "; VOG_LSTM_HALF_PROJ=$h python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "half_proj $h cfg2 1 stream: "; VOG_LSTM_HALF_PROJ=$h python bench.py $A --steps 1000 --warmup 40 --streams 1 2>/dev/null | tail -1
done; done
