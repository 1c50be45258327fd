#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "golden or batched or group or stress or replay" 2>&1 | tail -2
for wl in cfg2 cfg2x4 cfg5; do echo -n "$wl: "; python3 bench.py --gpus 1 --workload $wl --steps 400 --warmup 40 --throughput-only 2>/dev/null | tail -1; done
bash scratch/kt_forward.sh cfg2x4 20 1 2>&1 | grep -i "lstm"
