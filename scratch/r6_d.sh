#!/bin/bash
# round 6, call D: full GPU suite after the prune (with durations), CPU-independent smoke, bench K=20
R=$PWD; O=$R/gpurun_out/r6d; mkdir -p $O
python -m pytest tests -m gpu -q --durations=30 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -v "^$" $O/gpu_tests.log | tail -50
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 300 $O/bench_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6d/bench_k20.json').read().strip().splitlines()[-1])
print("value", d["value"], "steady", d.get("steady_state_400_steps",{}).get("value"), "f16", d.get("f16_transformers",{}).get("value"), "hi_lo", (d.get("hi_lo_plan_sharp16") or {}).get("value"), "b4", (d.get("requests_batched4") or {}).get("value"))
print({k:v for k,v in d["kernels_usec"].items() if v})
PY
