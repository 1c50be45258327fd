#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "argvec or srl" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_forward.py -m gpu -q -x -k "full_vs_reference or small" 2>&1 | tail -2
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
print(" ".join(f"{k} {eng.time_kernel(slot, k, 100):.2f}" for k in ("argvec", "mul_pl", "prep")))
PY
python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do echo "new $($B 2>/dev/null | cut -c1-120)"; done
