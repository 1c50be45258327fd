#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_split.py -m gpu -q -x 2>&1 | tail -3
bash scratch/r6_s.sh 2>&1 | head -1
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do echo "new $($B 2>/dev/null | cut -c1-120)"; done
for w in cfg3 cfg4 cfg5; do echo "$w $($B --workload $w 2>/dev/null | cut -c1-120)"; done
