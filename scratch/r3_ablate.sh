#!/bin/bash
# marginal cost of every launch of the cfg-2 forward with 4 forwards in flight (results are WRONG while skipping)
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r3abl; mkdir -p $O
for s in 1 4; do r=$(python bench.py --steps 800 --warmup 80 --streams $s --throughput-only 2>/dev/null | tail -1); echo "baseline streams=$s -> $r"; done > $O/ablate.log
for skip in prep "lstm_layer+vis_enc" "lstm_layer+obj_tail" "lstm_layer+vis_enc,lstm_layer+obj_tail" obj_qkv obj_attn "lstm_outproj+mul_pv" argvec mul_pl mul_attn mul_tail pred_head "argvec,mul_pl" "obj_qkv,obj_attn"; do
  r=$(VOG_SKIP_STEPS="$skip" python bench.py --steps 800 --warmup 80 --throughput-only 2>/dev/null | tail -1)
  echo "skip=[$skip] -> $r"
done >> $O/ablate.log
cat $O/ablate.log
