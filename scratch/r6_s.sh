#!/bin/bash
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
print(" ".join(f"{k} {eng.time_kernel(slot, k, 100):.2f}" for k in ("obj_qkv", "mul_pv")))
PY
for d in 0 1 2 3 4 5 6 7; do echo "debug $d: $(VOG_PERF_EXPERIMENTS=1 VOG_GEMM_DEBUG=$d python /tmp/tk.py 2>/dev/null)"; done
