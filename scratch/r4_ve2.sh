#!/bin/bash
A="--workload cfg4 --steps 200 --warmup 20 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do
echo -n "KC 256 paired:   "; python bench.py $A 2>/dev/null | tail -1
echo -n "KC 256 unpaired: "; python bench.py $A --set pair_mask=6 2>/dev/null | tail -1
echo -n "KC 128 paired:   "; VOG_HIP_LIB=/root/repo/scratch/tmp/kc128/libvog_hip.so python bench.py $A 2>/dev/null | tail -1
echo -n "KC 128 unpaired: "; VOG_HIP_LIB=/root/repo/scratch/tmp/kc128/libvog_hip.so python bench.py $A --set pair_mask=6 2>/dev/null | tail -1
done
A2="--steps 2000 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
echo -n "cfg2 KC 256: "; python bench.py $A2 2>/dev/null | tail -1
echo -n "cfg2 KC 128: "; VOG_HIP_LIB=/root/repo/scratch/tmp/kc128/libvog_hip.so python bench.py $A2 2>/dev/null | tail -1
VOG_HIP_LIB=/root/repo/scratch/tmp/kc128/libvog_hip.so bash scratch/kt_forward.sh cfg4 6 0 2>&1 | grep "vis_enc\|sum"
VOG_HIP_LIB=/root/repo/scratch/tmp/kc128/libvog_hip.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "enc" 2>&1 | tail -2
