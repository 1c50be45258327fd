#!/bin/bash
# hi + lo plan: stream-form encoders with 64-column workgroups
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -q -x -k "split or hi_lo or vis_enc or guard" 2>&1 | tail -3
bash scratch/r6_ab.sh 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null > /tmp/b.json
python - <<'PY'
import json
d = json.load(open("/tmp/b.json"))
print("value", d["value"], "hi_lo", d["hi_lo_plan_sharp16"]["value"], d["hi_lo_plan_sharp16"]["parity"]["rel_err_mdl_outs_eval"])
PY
done
for w in cfg3 cfg5; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > /tmp/b.json; python -c "
import json; d=json.load(open('/tmp/b.json')); print('$w', d['value'], 'hi_lo', d['hi_lo_plan_sharp16']['value'])"; done
