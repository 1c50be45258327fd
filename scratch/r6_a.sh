#!/bin/bash
# round 6, call A: GPU suite + A/B of the small-launch changes (argvec 16 wgs, mul_pl NT=4, prep fill 16/thread) + qkv_lean
R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
B="python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
for i in 1 2; do
  echo "default      $($B 2>/dev/null)"
  echo "nt4_off      $(VOG_PERF_EXPERIMENTS=1 VOG_SKINNY_NT4_OFF=1 $B 2>/dev/null)"
  echo "prep_fill1   $(VOG_PERF_EXPERIMENTS=1 VOG_PREP_FILL=1 $B 2>/dev/null)"
  echo "both_off     $(VOG_PERF_EXPERIMENTS=1 VOG_PREP_FILL=1 VOG_SKINNY_NT4_OFF=1 $B 2>/dev/null)"
  echo "qkv_lean1    $($B --set qkv_lean=1 2>/dev/null)"
done 2>&1 | tee $O/ab.txt
python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; tail -c 600 $O/bench_k20.err
bash scratch/prof_cu.sh 1 r6a > $O/busy_cu.txt 2>&1; cat $O/busy_cu.txt
