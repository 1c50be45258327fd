#!/bin/bash
# round 5: mul tail - stage-boundary weight priming (VOG_TAIL_PRIME) and placement on fewer XCDs (VOG_TAIL_XCDS)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
for r in 1 2; do
for v in base prime; do L=""; [ $v != base ] && L=/root/repo/scratch/tmp/$v/libvog_hip.so
for x in 0 4 2; do
echo -n "$v xcds=$x kernels: "; VOG_TAIL_XCDS=$x VOG_HIP_LIB=$L python scratch/mb_tail.py mul_tail obj_tail 2>/dev/null | tail -1
echo -n "$v xcds=$x cfg2 4 streams: "; VOG_TAIL_XCDS=$x VOG_HIP_LIB=$L python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
done; done; done
echo -n "cfg4 base: "; python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
echo -n "cfg4 prime: "; VOG_HIP_LIB=/root/repo/scratch/tmp/prime/libvog_hip.so python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
