#!/bin/bash
# QKV GEMM ring depth at cfg 2 (VOG_GEMM_TILE: 5 = 64x64 2 stages (shipped), 7 = 3 stages, 2 = 4 stages)
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
for k in ("obj_qkv", "mul_pv", "mul_pl", "lstm_outproj", "argvec", "prep", "pred"):
    try: print(k, round(eng.time_kernel(slot, k, 100), 2))
    except Exception as e: print(k, "n/a", str(e)[:60])
PY
for t in 5 7 2; do echo "tile $t:"; VOG_PERF_EXPERIMENTS=1 VOG_GEMM_TILE=$t python /tmp/tk.py 2>/dev/null; done
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2; do
  for t in 5 7 2; do echo "tile $t $(VOG_PERF_EXPERIMENTS=1 VOG_GEMM_TILE=$t $B 2>/dev/null | cut -c1-120)"; done
done
