#!/bin/bash
run() { python3 bench.py --gpus 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(d['ms_per_step']*1e3,2))"; }
run --steps 20 --warmup 5
run --steps 20 --warmup 200
run --steps 20 --warmup 2000
run --steps 40 --warmup 5
run --steps 40 --warmup 200
run --steps 100 --warmup 200
run --steps 20 --warmup 5 --streams 3
run --steps 20 --warmup 5 --streams 2
