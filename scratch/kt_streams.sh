R=$PWD; O=$R/gpurun_out/final2; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for s in 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_s$s -o r -- python $R/bench.py --throughput-only --streams $s --steps 400 --warmup 40 > $O/kt_s$s.log 2>&1
  db=$(find $O/kt_s$s -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_${s}streams.md 2>&1
done
cd $R; rm -rf $O/kt_s4 $O/kt_s1; tail -4 $O/kernel_stats_4streams.md; tail -3 $O/kernel_stats_1streams.md
