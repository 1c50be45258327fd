#!/bin/bash
# round 5: stream-form encoders INSIDE the BiLSTM layer-0 pair launch (one workgroup per CU there): pipeline depth 2 / 4 / 6
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
export VOG_PERF_EXPERIMENTS=1
for r in 1 2; do
for v in lean pd2 pd4 pd6; do L=""; S=1; [ $v = lean ] && S=0; [ $v = pd2 ] && L=/root/repo/scratch/tmp/pd2/libvog_hip.so; [ $v = pd6 ] && L=/root/repo/scratch/tmp/pd6/libvog_hip.so
echo -n "$v cfg2 kernels: "; VOG_VE_STREAM=$S VOG_HIP_LIB=$L python scratch/mb_tail.py "lstm_layer+vis_enc" vis_enc 2>/dev/null | tail -1
echo -n "$v cfg2: "; VOG_VE_STREAM=$S VOG_HIP_LIB=$L python bench.py $A --steps 2000 --warmup 40 2>/dev/null | tail -1
echo -n "$v cfg3: "; VOG_VE_STREAM=$S VOG_HIP_LIB=$L python bench.py $A --workload cfg3 --steps 1000 --warmup 40 2>/dev/null | tail -1
echo -n "$v cfg4 kernels: "; WL=cfg4 VOG_VE_STREAM=$S VOG_HIP_LIB=$L python scratch/mb_tail.py "lstm_layer+vis_enc" vis_enc 2>/dev/null | tail -1
echo -n "$v cfg4: "; VOG_VE_STREAM=$S VOG_HIP_LIB=$L python bench.py $A --workload cfg4 --steps 200 --warmup 20 2>/dev/null | tail -1
done; done
