#!/bin/bash
# round 5: stream-form encoders at p100 (VOG_VE_STREAM=0: round 4's lean form), DEPTH 2 / 3 / 4 chunks in flight
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
python -m pytest tests/test_gpu_forward.py -q -m gpu -x -k "p100" 2>&1 | tail -2
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "vis_enc" 2>&1 | tail -2
for r in 1 2; do
for v in lean vs2 vs3 vs4; do L=""; S=1; [ $v = lean ] && S=0; [ $v = vs2 ] && L=/root/repo/scratch/tmp/vs2/libvog_hip.so; [ $v = vs4 ] && L=/root/repo/scratch/tmp/vs4/libvog_hip.so
echo -n "$v kernels: "; WL=cfg4 VOG_VE_STREAM=$S VOG_HIP_LIB=$L python scratch/mb_tail.py vis_enc obj_qkv mul_pv 2>/dev/null | tail -1
echo -n "$v cfg4: "; VOG_VE_STREAM=$S VOG_HIP_LIB=$L python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done; done
