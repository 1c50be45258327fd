#!/bin/bash
# Round-6 evidence (run through gpurun from the repo root): bench lines for every workload (the driver's command for cfg 2),
# kernel-trace stats with 1 and 4 streams (+ every step as its own kernel), PMC traffic + MFMA utilisation for cfg 2, 3, 4, 5
# (separate passes: --pmc with --kernel-trace only), traffic of the default paired forward, cfg-4 kernel times, busy-CU table.
# PART=bench|kt|pmc|rest selects a part (gpurun calls are bounded); default: everything.
R=$PWD; O=$R/gpurun_out/final6; mkdir -p $O
PART=${PART:-all}
if [ $PART = all -o $PART = bench ]; then
python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_driver_steps20.json 2> $O/bench_cfg2_driver.err
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for wl in cfg3 cfg4 cfg5; do timeout 400 python bench.py --workload $wl --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_$wl.json 2> $O/bench_$wl.err; done
fi
export TMPDIR=/tmp; cd /tmp
if [ $PART = all -o $PART = kt ]; then
for s in 4 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_s$s -o r -- python $R/bench.py --throughput-only --streams $s --steps 400 --warmup 40 > $O/kt_s$s.log 2>&1
  db=$(find $O/kt_s$s -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_${s}streams.md 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_unp -o r -- python $R/scratch/prof_forward.py cfg2 200 0 > $O/kt_unp.log 2>&1
db=$(find $O/kt_unp -name "*.db" | head -1); [ -n "$db" ] && python $R/scratch/prof_summary.py $db > $O/kernel_stats_unpaired_1stream.md 2>&1
rm -rf $O/kt_s4 $O/kt_s1 $O/kt_unp
fi
if [ $PART = all -o $PART = pmc ]; then
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
for wl in ${WLS:-cfg2 cfg3 cfg4 cfg5}; do
  N=20; [ $wl = cfg4 ] && N=6
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_f -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_w -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_w.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_${wl}_sq -o r -- python $R/scratch/prof_forward.py $wl $N > $O/pmc_${wl}_sq.log 2>&1
  fc=$(find $O/pmc_${wl}_f -name "*counter_collection.csv" | head -1); wc=$(find $O/pmc_${wl}_w -name "*counter_collection.csv" | head -1); sc=$(find $O/pmc_${wl}_sq -name "*counter_collection.csv" | head -1)
  python $R/scratch/pmc_round2.py $wl $N "${fc:--}" "${wc:--}" "${sc:--}" $O/pmc_$wl.md $O/pmc_traffic.json profiles/round6_pmc_$wl.md > /dev/null 2>$O/pmc_${wl}_sum.err
  rm -rf $O/pmc_${wl}_f $O/pmc_${wl}_w $O/pmc_${wl}_sq
done
fi
cd $R
if [ $PART = all -o $PART = rest ]; then
bash scratch/traffic_total.sh > $O/traffic_default_forward.txt 2>&1
bash scratch/kt_forward.sh cfg4 6 0 > $O/kernel_times_cfg4.txt 2>&1
bash scratch/prof_cu.sh 1 fin > $O/busy_cu_cfg2.txt 2>&1
fi
ls $O | head -40
for wl in cfg2_driver_steps20 cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d=json.load(open("$O/bench_$wl.json")); print("$wl", d["value"] and round(d["value"],1), round(d["ms_per_step"]*1e3,2), "us", d["parity"]["ok"], d["parity"]["rel_err_mdl_outs_eval"], (d.get("steady_state_400_steps") or {}).get("value"), (d.get("f16_transformers") or {}).get("value"))
except Exception as e: print("$wl failed", e)
PY
done
