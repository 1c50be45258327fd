#!/bin/bash
python3 bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-train-extra 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K400', round(d['value']), d['parity']['ok'], round(d['roofline']['frac'],4), {k:v for k,v in d['kernels_usec'].items() if v and 'lstm_layer' in k})"
