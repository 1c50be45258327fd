#!/bin/bash
# experiment 5: true (saturated) CU cost of every launch: the ablation of round 3 repeated on HALF the chip (HSA_CU_MASK=0:0-127),
# where the 4-stream loop is CU-bound for certain (exp. 3c: time x 1.86 on half the CUs): marginal us per batch x 128 = CU-us
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4abl128; mkdir -p $O
run() { HSA_CU_MASK=0:0-127 timeout 200 python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>/dev/null | tail -1; }
{
echo "baseline -> $(run)"
for skip in prep "lstm_layer+vis_enc" "lstm_layer+obj_tail" obj_qkv obj_attn "obj_qkv,obj_attn" "lstm_outproj+mul_pv" argvec mul_pl mul_attn mul_tail pred_head; do
  echo "skip=[$skip] -> $(VOG_SKIP_STEPS="$skip" run)"
done
echo "qkv_lean=1 -> $(run --set qkv_lean=1)"
echo "pair_launches=0 -> $(run --set pair_launches=0)"
echo "pair_launches=0 enc_lean=1 -> $(run --set pair_launches=0 --set enc_lean=1)"
echo "fused_pred=1 -> $(run --set fused_pred=1)"
echo "baseline -> $(run)"
} > $O/ablate128.log 2>&1
cat $O/ablate128.log
