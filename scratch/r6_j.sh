#!/bin/bash
R=$PWD; O=$R/gpurun_out/r6j; mkdir -p $O
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for name in ("full/cfg2_vog_spat_gt5_bs4", "full/cfg4_vog_spat_p100_bs4"):
    try:
        eng, cfg, sd, batch, c, dev = build_engine(name, "bf16")
        slot = eng.make_slot(dev, graph=False)
        for k in ("obj_qkv", "obj_attn", "obj_tail", "mul_qkv", "mul_attn", "mul_tail", "pred_head", "argvec"):
            try: print(name[5:9], k, round(eng.time_kernel(slot, k, 50), 2))
            except Exception as e: print(name[5:9], k, "n/a", str(e)[:60])
    except Exception as e:
        print(name, "failed", e)
PY
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -v "^$" $O/gpu_tests.log | tail -8
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2 3; do
  echo "all   $($B 2>/dev/null)"
  echo "none  $(VOG_HIP_LIB=$R/scratch/tmp/nolm/libvog_hip.so $B 2>/dev/null)"
done
for w in cfg3 cfg4 cfg5; do echo "$w $($B --workload $w --steps 100 --warmup 10 2>/dev/null)"; done
