#!/bin/bash
export VOG_PERF_EXPERIMENTS=1
R=$PWD; O=$R/gpurun_out/r4ef3; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -x -q -k "struct" 2>&1 | tail -3 > $O/tests_ops.log
run() { timeout 300 python bench.py --steps 400 --warmup 40 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "cfg4: EF v2 deep prefetch, 1 wg/CU  -> $(run --workload cfg4)"
echo "cfg4: EF v2, 2 wg/CU                -> $(VOG_ATTN_STRUCT_EF=2 run --workload cfg4)"
done
} > $O/ef.log 2>&1
bash scratch/kt_forward.sh cfg4 6 0 > $O/kt_cfg4.txt 2>&1
cd /tmp
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -o r -- python $R/scratch/prof_forward.py cfg4 3 0 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scratch/pmc_kernels.py $f attn_struct_ef attn_tile2 >> $O/pmc.txt 2>&1
done
cd $R
find $O -name "*.csv" -size +5M -delete
cat $O/tests_ops.log $O/ef.log; grep -E "struct_ef|sum" $O/kt_cfg4.txt; cat $O/pmc.txt
