#!/bin/bash
# round 5: p100 QKV - all-columns row-block form (VOG_QKV_ROWALL=0: round 4's 256-column row blocks)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "p100" 2>&1 | tail -2
for r in 1 2; do
for v in 0 1; do
echo -n "rowall $v kernels: "; WL=cfg4 VOG_QKV_ROWALL=$v python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
echo -n "rowall $v cfg4: "; VOG_QKV_ROWALL=$v python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done; done
