#!/bin/bash
# round 5: helper workgroups for the second K half of the layer-1 input projection (lstm_helpers = 1)
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
python - <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, ".")
from oracle import cases
from tests.gpu_util import build_engine
from tests.test_gpu_forward import _check_against
for name in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg2_ragged", "full/cfg5_vog_svsq_gt5_bs16", "full/cfg2_sharp8"):
    eng, cfg, sd, batch, c, dev = build_engine(name)
    a = eng.forward(dev); torch.cuda.synchronize()
    eng.set_option("lstm_helpers", 1)
    for it in range(3):
        b = eng.forward(dev); torch.cuda.synchronize()
    eng.check()
    d = (a["mdl_outs"] - b["mdl_outs"]).abs().max().item()
    pred = eng.unpack_pred(b["pred_rec"], batch["new_srl_idxs"].shape[1])
    g = np.load(cases.golden_path(name))
    _check_against(name, b, pred, g, None, tol_rel=1e-3, tol_logit=6e-3)
    print(name, "helpers vs plain max abs logit diff", d)
PY
for r in 1 2 3; do
for h in 0 1; do
echo -n "helpers $h kernels: "; OPTS="lstm_helpers=$h" python scratch/mb_tail.py "lstm_layer+obj_tail" "lstm_layer#1" "lstm_layer#0" 2>/dev/null | tail -1
echo -n "helpers $h cfg2 4 streams: "; python bench.py $A --steps 2000 --warmup 40 --set lstm_helpers=$h 2>/dev/null | tail -1
echo -n "helpers $h cfg2 1 stream: "; python bench.py $A --steps 1000 --warmup 40 --streams 1 --set lstm_helpers=$h 2>/dev/null | tail -1
done; done
