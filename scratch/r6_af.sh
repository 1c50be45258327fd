#!/bin/bash
# hi + lo plan: obj_tx attention on the 8-wave form against the lean form, in-run
for i in 1 2; do for f in 1 0; do
VOG_PERF_EXPERIMENTS=1 VOG_ATTN_FRAG8=$f python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null > /tmp/b.json
python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("frag8=$f", "hi_lo", d["hi_lo_plan_sharp16"]["value"], "f16", (d.get("f16_transformers") or {}).get("value"))
PY
done; done
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for w in cfg3 cfg5; do echo "$w $($B --workload $w 2>/dev/null)"; done
