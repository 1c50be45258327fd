#!/bin/bash
# mul tail FFN1 stage: blocks 8-11 shared by two waves (balanced) against the two-pass form (scratch/tmp/ffn1_old)
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "tail" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_forward.py -m gpu -q -x -k "full_vs_reference or bf16_vs_reference" 2>&1 | tail -2
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
import os
for case in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg4_vog_spat_p100_bs4"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "bf16")
    slot = eng.make_slot(dev, graph=False)
    print(case, " ".join(f"{k} {eng.time_kernel(slot, k, 50):.2f}" for k in ("mul_tail", "obj_tail")))
PY
echo "balanced:"; python /tmp/tk.py 2>/dev/null
echo "old:"; VOG_HIP_LIB=$PWD/scratch/tmp/ffn1_old/libvog_hip.so python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do
  echo "balanced $($B 2>/dev/null | cut -c1-120)"
  echo "old      $(VOG_HIP_LIB=$PWD/scratch/tmp/ffn1_old/libvog_hip.so $B 2>/dev/null | cut -c1-120)"
done
for i in 1 2; do
  echo "cfg4 balanced $($B --workload cfg4 --steps 100 --warmup 10 2>/dev/null | cut -c1-120)"
  echo "cfg4 old      $(VOG_HIP_LIB=$PWD/scratch/tmp/ffn1_old/libvog_hip.so $B --workload cfg4 --steps 100 --warmup 10 2>/dev/null | cut -c1-120)"
done
