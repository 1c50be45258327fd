#!/bin/bash
# attn_frag8 (obj_tx attention, one round trip) against the lean form: op tests, goldens, kernel time, in-run A/B
R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -q -x -k "full_vs_reference or bf16_vs_reference or small" 2>&1 | tail -3
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
for k in ("mul_attn", "obj_attn"):
    print(k, round(eng.time_kernel(slot, k, 100), 2))
PY
echo "frag8:"; python /tmp/tk.py
echo "lean:"; VOG_PERF_EXPERIMENTS=1 VOG_ATTN_FRAG8=0 python /tmp/tk.py
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do
  echo "frag8 $($B 2>/dev/null | cut -c1-120)"
  echo "lean  $(VOG_PERF_EXPERIMENTS=1 VOG_ATTN_FRAG8=0 $B 2>/dev/null | cut -c1-120)"
done
for w in cfg3 cfg5; do
  echo "$w frag8 $($B --workload $w 2>/dev/null | cut -c1-120)"
  echo "$w lean  $(VOG_PERF_EXPERIMENTS=1 VOG_ATTN_FRAG8=0 $B --workload $w 2>/dev/null | cut -c1-120)"
done
