#!/bin/bash
# round 6, call B: hi + lo operand kernels - op tests, forward goldens, logit guard; then the bench line
R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O
python -m pytest tests/test_gpu_split.py -x -q -s > $O/split_ops.log 2>&1; echo "rc=$?" >> $O/split_ops.log; tail -15 $O/split_ops.log
python -m pytest tests/test_gpu_forward.py -q -s -k "hi_lo or logit or precision_plan or p100_bf16" > $O/split_fwd.log 2>&1; echo "rc=$?" >> $O/split_fwd.log
grep -v "^$" $O/split_fwd.log | grep -i "eval rel\|plan\|passed\|failed\|error\|rc=\|logit\|flips" | tail -80
python -m pytest tests/test_gpu_ops.py -q -k "argvec" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-train-extra > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 400 $O/bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6b/bench_k20.json').read().strip().splitlines()[-1])
print("value", d["value"], "steady", d.get("steady_state_400_steps",{}).get("value"), "f16", d.get("f16_transformers",{}).get("value"))
print("hi_lo", json.dumps(d.get("hi_lo_plan_sharp16"))[:600])
print("parity", d["parity"])
print("roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,str) or len(v)<40})
print({k:d["kernels_usec"][k] for k in ("argvec","mul_pl","prep")})
PY
