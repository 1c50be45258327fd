#!/bin/bash
# round 5: p100 encoders, wide stream form (128 rows x 256 columns per workgroup; VOG_VE_WIDE=0: 64 x 128)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "vis_encode" 2>&1 | tail -2
python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "p100" 2>&1 | tail -2
for r in 1 2 3; do
for v in 0 1; do
echo -n "wide $v kernels: "; WL=cfg4 VOG_VE_WIDE=$v python scratch/mb_tail.py "lstm_layer+vis_enc" vis_enc 2>/dev/null | tail -1
echo -n "wide $v cfg4: "; VOG_VE_WIDE=$v python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done; done
