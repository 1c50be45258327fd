#!/bin/bash
# layer-1 input projection in the layer kernel beyond 80 columns (big prologue, gates through memory) vs the GEMM launch (fused_ih = 6)
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -q -x -k "full_vs_reference or small or ragged or lstm or fused or gate_table or group or batched" 2>&1 | tail -3
cat > /tmp/tk.py <<'PY'
import sys, os
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg3_vog_temp_gt5_bs8", "full/cfg5_vog_svsq_gt5_bs16"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "f16" if "cfg5" in case else "bf16")
    if os.environ.get("FIH"): eng.set_option("fused_ih", int(os.environ["FIH"]))
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ("lstm_ih1", "lstm_layer#0", "lstm_layer#1", "lstm_layer+obj_tail"):
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, " ".join(out))
PY
echo "default (big prologue):"; python /tmp/tk.py 2>/dev/null
echo "fused_ih=6 (GEMM launch):"; FIH=6 python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for w in cfg3 cfg5; do for i in 1 2; do
  echo "$w big   $($B --workload $w 2>/dev/null)"
  echo "$w gemm  $($B --workload $w --set fused_ih=6 2>/dev/null)"
done; done
echo "cfg2 $($B 2>/dev/null)"
