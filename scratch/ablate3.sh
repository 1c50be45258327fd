#!/bin/bash
# marginal cost per step group, current defaults (persistent LSTM), 4 hipGraph streams
export VOG_PERF_EXPERIMENTS=1
base=$(python bench.py --steps 800 --warmup 80 --throughput-only 2>/dev/null | tail -1)
echo "baseline -> $base"
for skip in lstm_layer lstm_ih lstm_outproj,argvec,mul_pl lang_prep,vis_prep prop_enc,seg_enc,enc_finish obj_qkv,obj_wo,obj_ffn obj_attn obj_ln mul_pv mul_attn mul_wo mul_ln mul_ffn lin2 score,pred_head; do
  r=$(VOG_SKIP_STEPS="$skip" python bench.py --steps 800 --warmup 80 --throughput-only 2>/dev/null | tail -1)
  echo "skip=[$skip] -> $r"
done
