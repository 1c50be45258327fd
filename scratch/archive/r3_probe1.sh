#!/bin/bash
# round 3, probe 1: hand-off placement micro-benchmark + marginal cost of steps under the round-2 defaults
O=gpurun_out/r3p1; mkdir -p $O
timeout 120 scratch/ubench/xcd_allgather > $O/xcd_allgather.log 2>&1
export VOG_PERF_EXPERIMENTS=1
for s in 1 2 4; do
  r=$(python bench.py --steps 800 --warmup 80 --streams $s --throughput-only 2>/dev/null | tail -1); echo "baseline streams=$s -> $r"
done > $O/ablate.log
for skip in argvec pred_head obj_qkv,obj_attn "lstm_layer+vis_enc,lstm_layer+obj_tail" mul_attn mul_tail "lstm_outproj+mul_pv" prep; do
  r=$(VOG_SKIP_STEPS="$skip" python bench.py --steps 800 --warmup 80 --throughput-only 2>/dev/null | tail -1)
  echo "skip=[$skip] -> $r"
done >> $O/ablate.log
cat $O/xcd_allgather.log $O/ablate.log
