#!/bin/bash
# experiment 4: do two persistent BiLSTM layer kernels of different streams share CUs when their LDS allows it (78.5 KB each)?
# only prep + the two layer launches per forward (everything else skipped: results are WRONG), unpaired, on 64 / 128 CUs
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4coloc; mkdir -p $O
SK="vis_enc,obj_qkv,obj_attn,obj_tail,lstm_outproj,mul_pv,argvec,mul_pl,mul_attn,mul_tail,pred_head"
run() { timeout 200 python bench.py --steps 400 --warmup 40 --throughput-only --set pair_launches=0 "$@" 2>&1 | tail -1; }
{
for cus in 63 127 255; do
for s in 1 4; do
echo "lstm only, CUs 0-$cus, streams=$s, lds 78.5K -> $(HSA_CU_MASK=0:0-$cus VOG_SKIP_STEPS=$SK run --streams $s)"
echo "lstm only, CUs 0-$cus, streams=$s, lds 156K  -> $(HSA_CU_MASK=0:0-$cus VOG_SKIP_STEPS=$SK VOG_LSTM_LDS_EXTRA=79000 run --streams $s)"
done
done
} > $O/coloc.log 2>&1
cat $O/coloc.log
