#!/bin/bash
O=gpurun_out/r4eval2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_dist.py tests/test_gpu_forward.py -x -q -k "evaluator or main_dist or rank or batch_requests or Evaluator or prefetcher or fed_slot" 2>&1 | tail -8 > $O/tests_eval.log
for br in 1 4; do BR=$br timeout 300 python scratch/prof_eval_host.py > $O/prof_br$br.txt 2>&1; BR=$br NOPROF=1 timeout 300 python scratch/prof_eval_host.py > $O/noprof_br$br.txt 2>&1; done
grep -h "queries_per_s" $O/prof_br1.txt $O/noprof_br1.txt $O/prof_br4.txt $O/noprof_br4.txt | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['uid'], d['queries'], round(d['seconds'], 3), round(d['queries_per_s'], 1))
    except Exception: pass
"
A="--steps 400 --warmup 40 --no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0"
for w in cfg2 cfg4; do for via in zero_copy device; do
  VOG_BENCH_FED_VIA=$via timeout 300 python bench.py $A --workload $w > $O/$w.$via.json 2> $O/$w.$via.err
  python - <<PY
import json
d = json.loads(open("$O/$w.$via.json").read().strip().splitlines()[-1])
ba = d["batch_assembly"]
print("$w via=$via value", round(d["value"]), d["parity"]["ok"])
for k in ("measured_host_fed", "measured_host_fed_graph"):
    print("   ", k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in ba.get(k, {"missing": ba.get("error")}).items() if a != "what"})
PY
done; done
