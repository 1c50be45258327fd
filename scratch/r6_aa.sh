#!/bin/bash
# layer-0 gates by table look-up (fused_ih = 1) against the in-kernel / GEMM input projection (fused_ih = 4)
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -m gpu -q -x -k "full_vs_reference or small or ragged or lstm or fused or gate_table" 2>&1 | tail -3
cat > /tmp/tk.py <<'PY'
import sys, os
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg3_vog_temp_gt5_bs8", "full/cfg5_vog_svsq_gt5_bs16"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "f16" if "cfg5" in case else "bf16")
    if os.environ.get("FIH"): eng.set_option("fused_ih", int(os.environ["FIH"]))
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ("prep", "lstm_ih0", "lstm_layer#0", "lstm_layer#1"):
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, " ".join(out))
PY
echo "default:"; python /tmp/tk.py 2>/dev/null; echo "table (5):"; FIH=5 python /tmp/tk.py 2>/dev/null
echo "fused_ih=4:"; FIH=4 python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for w in cfg2 cfg3 cfg5; do for i in 1 2; do
  echo "$w table     $($B --workload $w 2>/dev/null)"
  echo "$w fused_ih4 $($B --workload $w --set fused_ih=4 2>/dev/null)"
done; done
