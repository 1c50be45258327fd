#!/bin/bash
# validation loop as a user runs it (main_dist --only_val, 128 synthetic loader batches): queries/s with 1 and 4 requests per forward
O=gpurun_out/r4eval; mkdir -p $O
python -m pytest tests/test_gpu_surface.py tests/test_gpu_dist.py -x -q -k "evaluator or main_dist or rank" 2>&1 | tail -3 > $O/tests.log
for br in 1 4; do BR=$br python scratch/prof_eval_host.py > $O/prof_br$br.txt 2>&1; done
grep -h "queries_per_s" $O/prof_br1.txt $O/prof_br4.txt | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['uid'], d['queries'], round(d['seconds'], 3), round(d['queries_per_s'], 1))
    except Exception: pass
"
cat $O/tests.log; head -40 $O/prof_br4.txt | tail -32
