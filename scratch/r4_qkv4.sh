#!/bin/bash
# cfg 4: forms of the QKV projections (row block 512 / 256 columns per workgroup, tiled LDS-DMA GEMM)
A="--workload cfg4 --steps 200 --warmup 20 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do
echo -n "row block 512 cols: "; python bench.py $A 2>/dev/null | tail -1
echo -n "row block 256 cols: "; VOG_PERF_EXPERIMENTS=1 VOG_QKV_NARROW=1 python bench.py $A 2>/dev/null | tail -1
echo -n "tiled gemm_pipe:    "; python bench.py $A --set qkv_lean=0 2>/dev/null | tail -1
done
VOG_PERF_EXPERIMENTS=1 VOG_QKV_NARROW=1 bash scratch/kt_forward.sh cfg4 6 0 2>&1 | grep -i "qkv\|sum"
