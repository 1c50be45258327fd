#!/bin/bash
# round 6, call F: remaining GPU tests after the fixes, then the round's evidence part 1 (bench lines)
R=$PWD; O=$R/gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -q -k "fp32_path_behind or argvec or slots_sharing or stress or determin" > $O/tests.log 2>&1; tail -3 $O/tests.log
PART=bench bash scratch/prof_round6.sh
for f in gpurun_out/final6/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d.get("steady_state_400_steps",{}).get("value"), d["parity"]["ok"], d["parity"]["rel_err_mdl_outs_eval"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), (d.get("hi_lo_plan_sharp16") or {}).get("value"), d["kernels_usec"].get("argvec"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
