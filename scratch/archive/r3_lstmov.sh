#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-train-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K400', d['value'], d['parity']['ok'], d['roofline']['frac'], d['roofline']['usec_per_launch'], {k:v for k,v in d['kernels_usec'].items() if v and 'lstm' in k})"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K20', d['value'], d['parity']['ok'], d['steady_state_400_steps']['value'])"
