#!/bin/bash
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/aqltrace -o r -- python $R/bench.py --steps 40 --warmup 8 --mode aql --queues 1 --interleave 4 --throughput-only > $R/gpurun_out/aqltrace.log 2>&1
tail -2 $R/gpurun_out/aqltrace.log
ls -la $R/gpurun_out/aqltrace
