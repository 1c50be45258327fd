#!/bin/bash
# weight prefetch depth of the tail's Wo / FFN2 stages at d = 768 (VOG_TAIL_PFA3: 4 shipped; variants under scratch/tmp/pfa3, pfa6)
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg4_vog_spat_p100_bs4"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "bf16")
    slot = eng.make_slot(dev, graph=False)
    print(case, f"mul_tail {eng.time_kernel(slot, 'mul_tail', 50):.2f}")
PY
for v in "" pfa3 pfa6; do
  L=""; [ -n "$v" ] && L="VOG_HIP_LIB=$PWD/scratch/tmp/$v/libvog_hip.so"
  echo "== ${v:-pf4}"; env $L python /tmp/tk.py 2>/dev/null
done
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2; do
  echo "pf4 $($B 2>/dev/null)"
  echo "pf3 $(VOG_HIP_LIB=$PWD/scratch/tmp/pfa3/libvog_hip.so $B 2>/dev/null)"
  echo "pf6 $(VOG_HIP_LIB=$PWD/scratch/tmp/pfa6/libvog_hip.so $B 2>/dev/null)"
done
