#!/bin/bash
mkdir -p gpurun_out/bwd2
timeout 900 python -m pytest tests/test_gpu_surface.py -x -q -m gpu -k "training or cli_fit" -s 2>&1 | tail -40 > gpurun_out/bwd2/test.log
cat gpurun_out/bwd2/test.log
