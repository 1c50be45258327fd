#!/bin/bash
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/full/gpu_tests.log
cat gpurun_out/full/gpu_tests.log
timeout 600 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
head -c 600 gpurun_out/full/bench.json; echo
