#!/bin/bash
# (layer-1 input projection GEMM || obj_tx QKV) as one launch at cfg 3 / cfg 5 (pair_mask 15) against apart (7)
timeout 1500 python -m pytest tests/test_gpu_forward.py -m gpu -q -x -k "full_vs_reference or pair or group or batched" 2>&1 | tail -3
cat > /tmp/tk.py <<'PY'
import sys, os
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg3_vog_temp_gt5_bs8", "full/cfg5_vog_svsq_gt5_bs16"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "f16" if "cfg5" in case else "bf16")
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ("lstm_ih1", "obj_qkv", "lstm_ih1+obj_qkv", "obj_attn"):
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, " ".join(out))
PY
python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for w in cfg3 cfg5; do for i in 1 2 3; do
  echo "$w pair15 $($B --workload $w 2>/dev/null)"
  echo "$w pair7  $($B --workload $w --set pair_mask=7 2>/dev/null)"
done; done
echo "cfg2 $($B 2>/dev/null)"
# obj_tx attention of the hi + lo plan on the 8-wave one-round-trip form
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -q -x -k "split or hi_lo or attention or guard" 2>&1 | tail -3
cat > /tmp/tk2.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_sharp16", "auto")
slot = eng.make_slot(dev, graph=False)
print(eng.plan, " ".join(f"{k} {eng.time_kernel(slot, k, 100):.2f}" for k in ("obj_attn", "mul_attn")))
PY
echo "frag8 split:"; python /tmp/tk2.py 2>/dev/null
echo "lean split:"; VOG_PERF_EXPERIMENTS=1 VOG_ATTN_FRAG8=0 python /tmp/tk2.py 2>/dev/null
python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null > /tmp/b.json
python - <<'PY'
import json
d = json.load(open("/tmp/b.json"))
print("value", d["value"], "hi_lo", d["hi_lo_plan_sharp16"]["value"], d["hi_lo_plan_sharp16"]["parity"])
PY
