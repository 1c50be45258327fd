#!/bin/bash
R=$PWD
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do
  echo "all   $($B 2>/dev/null)"
  echo "none  $(VOG_HIP_LIB=$R/scratch/tmp/nolm/libvog_hip.so $B 2>/dev/null)"
  echo "samp4 $(VOG_HIP_LIB=$R/scratch/tmp/lm4/libvog_hip.so $B 2>/dev/null)"
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
