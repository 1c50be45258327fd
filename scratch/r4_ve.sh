#!/bin/bash
O=gpurun_out/r4ve; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -k "enc or golden or pair or chain or long or p100 or tile2 or guard" 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
A="--no-train-extra --no-cpu-baseline --no-cobatch-extra --rotate-inputs 0 --throughput-only"
for r in 1 2; do
for w in cfg2 cfg3 cfg4 cfg5; do echo -n "$w: "; python bench.py $A --workload $w --steps $([ $w = cfg4 ] && echo 200 || echo 2000) --warmup 40 2>/dev/null | tail -1; done
done
echo -n "cfg2 1 stream: "; python bench.py $A --streams 1 --steps 1000 2>/dev/null | tail -1
bash scratch/kt_forward.sh cfg4 6 0 2>&1 | tail -20 > gpurun_out/r4ve/kt_cfg4.txt; tail -3 gpurun_out/r4ve/kt_cfg4.txt
