#!/bin/bash
O=gpurun_out/r4eval4; mkdir -p $O
BR=1 timeout 300 python scratch/prof_eval_host.py > $O/prof_br1.txt 2>&1
grep -n "Function.*called" -A60 $O/prof_br1.txt | cut -c1-200
