#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -m gpu -k "enc or golden or pair" 2>&1 | tail -2
for wl in cfg4 cfg2; do echo -n "$wl: "; python3 bench.py --gpus 1 --workload $wl --steps 200 --warmup 20 --throughput-only 2>/dev/null | tail -1; done
bash scratch/kt_forward.sh cfg4 6 0 2>/dev/null | grep -i "vis_enc\|seg_rep"
