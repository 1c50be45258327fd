#!/bin/bash
R=$PWD; O=$R/gpurun_out/final6; mkdir -p $O
( time python -m pytest tests -m gpu -q -x --durations=12 ) > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -E "passed|failed|rc=|real" $O/gpu_tests.log | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; python -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['parity']['ok'])"
