#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_surface.py -m gpu -q 2>&1 | tail -4
