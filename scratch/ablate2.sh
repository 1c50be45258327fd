#!/bin/bash
# marginal cost per step group, persistent-LSTM default, hipGraph streams
for s in 1 2 3 4; do echo -n "streams=$s: "; python bench.py --steps 400 --warmup 40 --streams $s --throughput-only 2>/dev/null | tail -1; done
for skip in lstm_layer lstm_ih lstm_outproj,argvec,mul_pl,lang_prep vis_prep,prop_enc,seg_enc,enc_finish obj_ mul_pv mul_attn mul_wo mul_ln mul_ffn lin2,score,pred_head mul_; do
  r=$(VOG_PERF_EXPERIMENTS=1 VOG_SKIP_STEPS="$skip" python bench.py --steps 400 --warmup 40 --streams 4 --throughput-only 2>/dev/null | tail -1)
  echo "skip=[$skip] streams=4 -> $r"
done
