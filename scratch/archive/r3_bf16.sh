#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_surface.py -x -q -m gpu -k "mixed_precision" -s 2>&1 | tail -5 | cut -c1-400
for b in 0 1; do TRAIN_BF16=$b timeout 300 python scratch/time_train.py 2>&1 | tail -1; done
TRAIN_DROPOUT=1 timeout 300 python scratch/time_train.py 2>&1 | tail -1
