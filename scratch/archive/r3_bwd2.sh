#!/bin/bash
mkdir -p gpurun_out/bwd2
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_dist.py -x -q -m gpu -k "backward or training or cli_fit or dropout or mixed or two_rank" 2>&1 | tail -4 > gpurun_out/bwd2/test.log
cat gpurun_out/bwd2/test.log | cut -c1-330
for b in 0 1; do TRAIN_BF16=$b timeout 300 python scratch/time_train.py 2>&1 | tail -1; done
