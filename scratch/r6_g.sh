#!/bin/bash
# round 6, call G: argvec re-check, then evidence parts kt + pmc (cfg2) + rest
R=$PWD; O=$R/gpurun_out/r6g; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -q -k "argvec or (golden and small) or stages" > $O/tests.log 2>&1; tail -2 $O/tests.log
python - <<'PY'
import importlib, torch, sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
eng, cfg, sd, batch, c, dev = build_engine("full/cfg2_vog_spat_gt5_bs4", "bf16")
slot = eng.make_slot(dev, graph=False)
for k in ("argvec", "mul_pl", "prep", "mul_attn", "obj_attn", "pred_head"):
    print(k, round(eng.time_kernel(slot, k, 100), 2))
PY
PART=kt bash scratch/prof_round6.sh > $O/kt.log 2>&1
WLS="cfg2" PART=pmc bash scratch/prof_round6.sh > $O/pmc.log 2>&1
PART=rest bash scratch/prof_round6.sh > $O/rest.log 2>&1
ls gpurun_out/final6; cat gpurun_out/final6/busy_cu_cfg2.txt; head -30 gpurun_out/final6/kernel_stats_4streams.md
