#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6h; mkdir -p $O
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
for k in ("argvec", "mul_pl", "prep", "mul_attn", "obj_attn", "pred_head"):
    print(k, round(eng.time_kernel(slot, k, 100), 2))
PY
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -v "^$" $O/gpu_tests.log | tail -16
grep "plan \|planned" $O/gpu_tests.log | head -20
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do echo "cfg2 $($B 2>/dev/null)"; done
echo "cfg4 $($B --workload cfg4 --steps 100 --warmup 10 2>/dev/null)"
