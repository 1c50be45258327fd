#!/bin/bash
# kernel times of the hi + lo plan (cfg 2, wq / wk x 16 checkpoint) beside the plain plan
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case, tx in (("full/cfg2_vog_spat_gt5_bs4", "auto"), ("full/cfg2_sharp16", "auto")):
    eng, cfg, sd, batch, c, dev = build_engine(case, tx)
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ("prep", "lstm_layer#0", "vis_enc", "lstm_layer+vis_enc", "obj_qkv", "obj_attn", "obj_tail", "lstm_layer#1", "lstm_layer+obj_tail", "lstm_outproj", "mul_pv", "lstm_outproj+mul_pv", "argvec", "mul_pl", "mul_attn", "mul_tail", "pred_head"):
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, eng.plan, "\n   ", " | ".join(out))
PY
python /tmp/tk.py 2>/dev/null
