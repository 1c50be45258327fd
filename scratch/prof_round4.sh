#!/bin/bash
# Round-4 evidence (run through gpurun from the repo root): bench lines for every workload, kernel-trace
# stats with 1 and 4 streams (+ every step as its own kernel: the stand-alone lstm_layer_kernel rows), PMC
# traffic + MFMA utilisation for cfg2 and cfg4, traffic of the default paired forward, busy-CU table.
# Every rocprofv3 run is bounded; counters are collected in their own passes (--pmc with --kernel-trace only).
R=$PWD; O=$R/gpurun_out/final4; mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for wl in cfg3 cfg4 cfg5; do timeout 300 python bench.py --workload $wl --no-cobatch-extra --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_$wl.json 2> $O/bench_$wl.err; done
export TMPDIR=/tmp; cd /tmp
for s in 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_s$s -o r -- python $R/bench.py --throughput-only --streams $s --steps 400 --warmup 40 > $O/kt_s$s.log 2>&1
  db=$(find $O/kt_s$s -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_${s}streams.md 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_unp -o r -- python $R/scratch/prof_forward.py cfg2 200 0 > $O/kt_unp.log 2>&1
db=$(find $O/kt_unp -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_unpaired_1stream.md 2>&1
for wl in cfg2 cfg4; do
  N=20; [ $wl = cfg4 ] && N=6
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_f -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_w -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_${wl}_sq -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_sq.log 2>&1
  fc=$(find $O/pmc_${wl}_f -name "*counter_collection.csv" | head -1); wc=$(find $O/pmc_${wl}_w -name "*counter_collection.csv" | head -1); sc=$(find $O/pmc_${wl}_sq -name "*counter_collection.csv" | head -1)
  python $R/scratch/pmc_round2.py $wl $N "${fc:--}" "${wc:--}" "${sc:--}" $O/pmc_$wl.md $O/pmc_traffic.json profiles/round4_pmc_$wl.md > /dev/null 2>$O/pmc_${wl}_sum.err
done
cd $R
rm -rf $O/pmc_*_f $O/pmc_*_w $O/pmc_*_sq $O/kt_s4 $O/kt_s1 $O/kt_unp
bash scratch/traffic_total.sh > $O/traffic_default_forward.txt 2>&1
bash scratch/kt_forward.sh cfg4 6 0 > $O/kernel_times_cfg4.txt 2>&1
bash scratch/prof_cu.sh 1 fin > $O/busy_cu_cfg2.txt 2>&1
ls -la $O; head -c 400 $O/bench_cfg2.json; echo; for wl in cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$wl.json")); print("$wl", d["value"] and round(d["value"],1), round(d["ms_per_step"]*1e3,2), "us", d["parity"]["ok"], d["parity"]["rel_err_mdl_outs_eval"])
except Exception as e: print("$wl failed", e)
PY
done
tail -3 $O/kernel_stats_4streams.md; grep lstm_layer_kernel $O/kernel_stats_unpaired_1stream.md
