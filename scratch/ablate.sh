#!/bin/bash
# Marginal cost of each step group in the throughput regime (results are wrong while skipping).
for skip in "" lstm_step lstm_ih lstm_outproj,argvec,mul_pl,lang_prep vis_prep,prop_enc,seg_enc,enc_finish obj_ obj_attn mul_pv mul_attn mul_wo mul_ln mul_ffn lin2,score,pred_head mul_ "obj_,mul_,lin2,score,pred_head" "lstm,argvec,mul_pl,lang_prep"; do
  for s in 1 4; do
    r=$(VOG_PERF_EXPERIMENTS=1 VOG_SKIP_STEPS="$skip" python bench.py --steps 400 --warmup 40 --streams $s --throughput-only 2>/dev/null | tail -1)
    echo "skip=[$skip] streams=$s -> $r"
  done
done
