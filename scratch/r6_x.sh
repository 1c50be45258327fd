#!/bin/bash
# ring depth of the LDS-DMA GEMM for the skinny-M, long-K BiLSTM input projections (cfg 3 / cfg 5)
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case, ks in (("full/cfg3_vog_temp_gt5_bs8", ("lstm_ih0", "lstm_ih1", "obj_qkv", "mul_pv")), ("full/cfg5_vog_svsq_gt5_bs16", ("lstm_ih0", "lstm_ih1", "obj_qkv", "mul_pv"))):
    eng, cfg, sd, batch, c, dev = build_engine(case, "f16" if "cfg5" in case else "bf16")
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ks:
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, " ".join(out))
PY
echo "default:"; python /tmp/tk.py 2>/dev/null
for t in 7 2 8 10 9; do echo "tile $t:"; VOG_PERF_EXPERIMENTS=1 VOG_GEMM_TILE=$t python /tmp/tk.py 2>/dev/null; done
