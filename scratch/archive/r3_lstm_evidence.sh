#!/bin/bash
# round-3 evidence of the persistent BiLSTM: phase timeline of the layer kernel, the hand-off protocol in
# isolation (scratch/ubench), and the N > 1 stand-in: the RCCL exchange path with one rank, 4 streams, 10 k steps
O=gpurun_out/r3lstm; mkdir -p $O
for f in 0 512 2048; do echo "== ts_layer FUSED=$f (K of the in-kernel input projection; 0 = gates from memory)"; FUSED=$f VERBOSE=1 timeout 60 scratch/ts_layer; done > $O/ts_layer.txt 2>&1
timeout 120 scratch/ubench/handoff_v3 > $O/handoff_v3.txt 2>&1
timeout 120 scratch/ubench/xcd_allgather > $O/xcd_allgather.txt 2>&1
VOG_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 10000 --warmup 400 --no-cpu-baseline --no-cobatch-extra > $O/force_dist_10k.json 2> $O/force_dist_10k.err
python - <<PY
import json
d = json.load(open("$O/force_dist_10k.json"))
print("force_dist 10k steps:", d["value"], d["ms_per_step"], d["parity"], d["rccl_ranks"])
PY
tail -3 $O/ts_layer.txt; tail -4 $O/handoff_v3.txt
