#!/bin/bash
O=gpurun_out/r4eval3; mkdir -p $O
for br in 1 4; do BR=$br timeout 300 python scratch/prof_eval_host.py > $O/prof_br$br.txt 2>&1; for r in 1 2; do BR=$br NOPROF=1 timeout 300 python scratch/prof_eval_host.py > $O/noprof_br$br.$r.txt 2>&1; done; done
grep -h "queries_per_s" $O/noprof_br1.*.txt $O/noprof_br4.*.txt | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['uid'], d['queries'], round(d['seconds'], 3), round(d['queries_per_s'], 1))
    except Exception: pass
"
