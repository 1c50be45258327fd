#!/bin/bash
# BiLSTM layer kernel at 196 registers (one hand-off chunk in flight, enough at Bn = 4) against 208 (four in flight)
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do
  echo "infl4 (208 regs) $($B 2>/dev/null)"
  echo "infl1 (196 regs) $(VOG_HIP_LIB=$PWD/scratch/tmp/infl1/libvog_hip.so $B 2>/dev/null)"
done
