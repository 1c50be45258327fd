#!/bin/bash
# round 4, experiment 2: where are the 59 us between the kernels of a forward under load? (a) per-boundary gaps by kernel
# pair, 1 vs 4 streams (kernel trace only); (b) the AQL path with the fences between a forward's kernels switched off
R=$PWD; O=$R/gpurun_out/r4gaps; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for s in 4 1; do
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/s$s -o r -- python $R/bench.py --throughput-only --streams $s --steps 400 --warmup 40 > $O/s$s.log 2>&1
  python $R/scratch/gap_analysis.py $O/s$s "streams=$s" > $O/gaps_s$s.md 2>&1
done
cd $R
export VOG_PERF_EXPERIMENTS=1
{
for f in 1 0 11 21; do
  r=$(VOG_AQL_FENCE=$f timeout 300 python bench.py --steps 480 --warmup 48 --mode aql --queues 4 --interleave 1 --throughput-only 2>&1 | tail -1)
  echo "aql fence=$f queues=4 -> $r"
done
echo "graph streams=4 -> $(python bench.py --steps 800 --warmup 80 --throughput-only 2>/dev/null | tail -1)"
echo "graph streams=4 rotate -> $(python bench.py --steps 800 --warmup 80 --throughput-only --rotate-main 2>/dev/null | tail -1)"
} > $O/aql_fence.log 2>&1
rm -rf $O/s4/*/*.db $O/s1/*/*.db 2>/dev/null
find $O -name "*.csv" -size +20M -delete
cat $O/gaps_s4.md $O/gaps_s1.md $O/aql_fence.log
