#!/bin/bash
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2 3; do
for v in base pf8; do L=""; [ $v != base ] && L=/root/repo/scratch/tmp/$v/libvog_hip.so
echo -n "$v cfg2 4 streams: "; VOG_HIP_LIB=$L python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "$v cfg2 1 stream:  "; VOG_HIP_LIB=$L python bench.py $A --steps 1000 --warmup 40 --streams 1 2>/dev/null | tail -1
echo -n "$v cfg4:           "; VOG_HIP_LIB=$L python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done; done
