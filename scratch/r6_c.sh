#!/bin/bash
# round 6, call C: BiLSTM hand-off fetch with 1 / 2 / 4 chunks in flight per thread (cfg 2 / 3 / 5), fused projection up to 80 columns,
# split op tests + the fixed forward tests, LSTM goldens
R=$PWD; O=$R/gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_split.py -q -s > $O/split_ops.log 2>&1; echo "rc=$?" >> $O/split_ops.log; grep -i "error vs\|passed\|failed\|rc=" $O/split_ops.log | tail -30
python -m pytest tests/test_gpu_forward.py -q -s -k "hi_lo or logit or precision_plan or lstm or ragged or golden" > $O/fwd.log 2>&1; echo "rc=$?" >> $O/fwd.log
grep -i "plan \|passed\|failed\|rc=\|planned\|Error" $O/fwd.log | tail -40
python -m pytest tests/test_gpu_surface.py -q -k "stall or stress or slot" 2>&1 | tail -3
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2; do
 for wl in cfg2 cfg3 cfg5; do
  echo "$wl infl4 $($B --workload $wl 2>/dev/null)"
  echo "$wl infl2 $(VOG_HIP_LIB=$R/scratch/tmp/infl2/libvog_hip.so $B --workload $wl 2>/dev/null)"
  echo "$wl infl1 $(VOG_HIP_LIB=$R/scratch/tmp/infl1/libvog_hip.so $B --workload $wl 2>/dev/null)"
 done
done 2>&1 | tee $O/ab_infl.txt
for wl in cfg3 cfg5; do python bench.py --workload $wl --steps 400 --warmup 40 --no-train-extra --no-cpu-baseline > $O/bench_$wl.json 2>/dev/null; python - $O/bench_$wl.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"], d["value"], d["parity"]["ok"], d["parity"]["rel_err_mdl_outs_eval"], {k:v for k,v in d["kernels_usec"].items() if v and k.startswith("lstm")}, (d.get("hi_lo_plan_sharp16") or {}).get("value"))
PY
done
