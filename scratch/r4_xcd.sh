#!/bin/bash
# BiLSTM layer launches pinned to an XCD pair per forward lane (hand-off inside one L2)
O=gpurun_out/r4xcd; mkdir -p $O
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for pin in 0 -1; do
  for s in 1 4; do for r in 1 2; do
    echo -n "pin=$pin streams=$s: "; timeout 200 python bench.py $A --steps 2000 --warmup 100 --streams $s --set lstm_xcd_pin=$pin 2>&1 | tail -1
  done; done
  echo -n "pin=$pin K=20: "; timeout 200 python bench.py $A --steps 20 --warmup 5 --set lstm_xcd_pin=$pin 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_gpu_forward.py -x -q -k "golden or handoff or replay" 2>&1 | tail -3
