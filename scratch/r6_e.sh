#!/bin/bash
# round 6, call E: re-check after the logit-max / argvec fixes + the full suite, then the bench lines
R=$PWD; O=$R/gpurun_out/r6e; mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=12 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -v "^$" $O/gpu_tests.log | tail -25
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 300 $O/bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e/bench_k20.json').read().strip().splitlines()[-1])
print("value", d["value"], "steady", d.get("steady_state_400_steps",{}).get("value"), "f16", d.get("f16_transformers",{}).get("value"), "hi_lo", (d.get("hi_lo_plan_sharp16") or {}).get("value"), "b4", (d.get("requests_batched4") or {}).get("value"))
print({k:v for k,v in d["kernels_usec"].items() if v})
PY
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do echo "cfg2 $($B 2>/dev/null)"; done
