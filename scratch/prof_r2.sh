#!/bin/bash
# Round-2 kernel-trace profiles: gpurun -- bash scratch/prof_r2.sh   (every rocprofv3 run is bounded)
R=$PWD; O=$R/gpurun_out/prof_r2; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for s in 4 1; do
  timeout 240 rocprofv3 --kernel-trace --stats -d $O/s$s -o r -- python $R/bench.py --no-cpu-baseline --no-cobatch-extra --streams $s --steps 200 --warmup 20 --kernel-iters 20 > $O/s$s.log 2>&1
  db=$(find $O/s$s -name "*.db" | head -1)
  [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_${s}streams.md 2>&1
done
cd $R; head -40 $O/kernel_stats_4streams.md; tail -4 $O/kernel_stats_4streams.md; tail -4 $O/kernel_stats_1streams.md
