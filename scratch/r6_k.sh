#!/bin/bash
R=$PWD; O=$R/gpurun_out/final6; mkdir -p $O
( time python -m pytest tests -m gpu -q -x --durations=0 ) > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -E "passed|failed|rc=|real" $O/gpu_tests.log | tail -5
PART=bench bash scratch/prof_round6.sh 2>&1 | tail -8
PART=kt bash scratch/prof_round6.sh 2>&1 | tail -3
PART=rest bash scratch/prof_round6.sh 2>&1 | tail -3
