#!/bin/bash
mkdir -p gpurun_out/bwd2
timeout 900 python -m pytest tests/test_gpu_surface.py -x -q -m gpu -k "backward or training or cli_fit or dropout" -s 2>&1 | tail -30 > gpurun_out/bwd2/test.log
cat gpurun_out/bwd2/test.log | cut -c1-330
