#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "chained_obj_qkv" 2>&1 | grep -E "^E|passed|failed|Error" | head -8 | cut -c1-300
for c in 0 1; do for s in 4 1; do echo -n "chain_obj_qkv=$c streams=$s: "; python3 bench.py --gpus 1 --steps 400 --warmup 40 --streams $s --throughput-only --set chain_obj_qkv=$c 2>/dev/null | tail -1; done; done
for c in 0 1 0 1; do echo -n "chain=$c K20: "; python3 bench.py --gpus 1 --steps 20 --warmup 5 --throughput-only --set chain_obj_qkv=$c 2>/dev/null | tail -1; done
