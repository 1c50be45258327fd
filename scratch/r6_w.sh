#!/bin/bash
# (1) argvec with padded slices; (2) column-major tile order for the skinny-M BiLSTM input projections (cfg 3 / cfg 5) vs scratch/tmp/rowmajor
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "argvec or srl or gemm" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_forward.py -m gpu -q -x -k "full_vs_reference or small" 2>&1 | tail -2
cat > /tmp/tk.py <<'PY'
import sys
sys.path.insert(0, ".")
from tests.gpu_util import build_engine
for case, ks in (("full/cfg2_vog_spat_gt5_bs4", ("argvec", "mul_pl")), ("full/cfg3_vog_temp_gt5_bs8", ("lstm_ih0", "lstm_ih1", "argvec")), ("full/cfg5_vog_svsq_gt5_bs16", ("lstm_ih0", "lstm_ih1", "argvec"))):
    eng, cfg, sd, batch, c, dev = build_engine(case, "f16" if "cfg5" in case else "bf16")
    slot = eng.make_slot(dev, graph=False)
    out = []
    for k in ks:
        try: out.append(f"{k} {eng.time_kernel(slot, k, 100):.2f}")
        except Exception as e: out.append(f"{k} n/a")
    print(case, " ".join(out))
PY
echo "new:"; python /tmp/tk.py 2>/dev/null
echo "rowmajor:"; VOG_HIP_LIB=$PWD/scratch/tmp/rowmajor/libvog_hip.so python /tmp/tk.py 2>/dev/null
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2; do echo "cfg2 new $($B 2>/dev/null | cut -c1-120)"; done
for w in cfg3 cfg5; do for i in 1 2; do
  echo "$w colmajor $($B --workload $w 2>/dev/null | cut -c1-120)"
  echo "$w rowmajor $(VOG_HIP_LIB=$PWD/scratch/tmp/rowmajor/libvog_hip.so $B --workload $w 2>/dev/null | cut -c1-120)"
done; done
