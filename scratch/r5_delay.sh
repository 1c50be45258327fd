#!/bin/bash
# round 5: does a late start of the partner body give the BiLSTM layer's prologue back its stand-alone speed?
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
for r in 1 2; do
for d in 0 4 8 12 20; do
echo -n "delay $d kernels: "; VOG_PAIR_DELAY=$d python scratch/mb_tail.py "lstm_layer+vis_enc" "lstm_layer+obj_tail" "lstm_layer#0" "lstm_layer#1" 2>/dev/null | tail -1
echo -n "delay $d cfg2: "; VOG_PAIR_DELAY=$d python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
done; done
