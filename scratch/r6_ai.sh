#!/bin/bash
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg4_vog_spat_p100_bs4"):
    eng, cfg, sd, batch, c, dev = build_engine(case, "bf16")
    slot = eng.make_slot(dev, graph=False)
    print(case, f"mul_tail {eng.time_kernel(slot, 'mul_tail', 50):.2f}")
PY
echo "== pf2"; VOG_HIP_LIB=$PWD/scratch/tmp/pfa2/libvog_hip.so python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 100 --warmup 10 --throughput-only --no-cpu-baseline --workload cfg4"
for i in 1 2 3; do
  echo "cfg4 pf4 $($B 2>/dev/null)"
  echo "cfg4 pf3 $(VOG_HIP_LIB=$PWD/scratch/tmp/pfa3/libvog_hip.so $B 2>/dev/null)"
  echo "cfg4 pf2 $(VOG_HIP_LIB=$PWD/scratch/tmp/pfa2/libvog_hip.so $B 2>/dev/null)"
done
