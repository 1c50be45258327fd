#!/bin/bash
# kernel-time breakdown of the device training step (scratch/time_train.py) -> gpurun_out/train_prof/kernel_stats.md
R=$PWD; O=$R/gpurun_out/train_prof; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/raw -o r -- python $R/scratch/time_train.py > $O/run.log 2>&1
db=$(find $O/raw -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats.md 2>&1
rm -rf $O/raw
head -30 $O/kernel_stats.md
