#!/bin/bash
# round 5: p100 - BiLSTM layer 1 inside obj_tx's long attention launch (pair_attn = 1) vs next to the obj tail (0)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "p100" 2>&1 | tail -2
for r in 1 2 3; do
for v in 0 1; do
echo -n "pair_attn $v cfg4 4 streams: "; python bench.py $A --workload cfg4 --steps 200 --warmup 20 --set pair_attn=$v 2>/dev/null | tail -1
echo -n "pair_attn $v cfg4 1 stream: "; python bench.py $A --workload cfg4 --steps 100 --warmup 10 --streams 1 --set pair_attn=$v 2>/dev/null | tail -1
done; done
