#!/bin/bash
# round 5: upper bound of "QKV inside the attention kernels at gt5": REMOVE the projection launches (wrong results) - a fused kernel
# keeps their work (weight stream + MFMAs) and loses only the launch boundary and the fragment round trip
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
for r in 1 2 3; do
for skip in "" obj_qkv mul_pl obj_qkv,mul_pl; do
echo -n "skip=[$skip] 4 streams: "; VOG_SKIP_STEPS="$skip" python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "skip=[$skip] 1 stream: "; VOG_SKIP_STEPS="$skip" python bench.py $A --steps 1000 --warmup 40 --streams 1 2>/dev/null | tail -1
done; done
