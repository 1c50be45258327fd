#!/bin/bash
# FFN1 weight prefetch depth of the tail (VOG_TAIL_PF1: 4 shipped) with the Wo / FFN2 stages at 3
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg4_vog_spat_p100_bs4"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "bf16")
    slot = eng.make_slot(dev, graph=False)
    print(case, " ".join(f"{k} {eng.time_kernel(slot, k, 50):.2f}" for k in ("mul_tail", "obj_tail")))
PY
for v in "" pf1_2 pf1_3 pf1_6; do
  L=""; [ -n "$v" ] && L="VOG_HIP_LIB=$PWD/scratch/tmp/$v/libvog_hip.so"
  echo "== ${v:-pf1_4}"; env $L python /tmp/tk.py 2>/dev/null
done
B="python bench.py --steps 100 --warmup 10 --throughput-only --no-cpu-baseline --workload cfg4"
for i in 1 2; do
  echo "cfg4 pf1_4 $($B 2>/dev/null)"
  echo "cfg4 pf1_3 $(VOG_HIP_LIB=$PWD/scratch/tmp/pf1_3/libvog_hip.so $B 2>/dev/null)"
  echo "cfg4 pf1_6 $(VOG_HIP_LIB=$PWD/scratch/tmp/pf1_6/libvog_hip.so $B 2>/dev/null)"
done
