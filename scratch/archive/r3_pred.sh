#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "pred_head_in_score_tail or reference_golden" 2>&1 | tail -5
for fp in 0 1; do
  python3 bench.py --gpus 1 --steps 400 --warmup 40 --throughput-only --set fused_pred=$fp 2>/dev/null | tail -1
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K20', d['value'], d['parity']['ok'], d['steady_state_400_steps']['value'])"
