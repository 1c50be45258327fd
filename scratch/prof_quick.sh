#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pq
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/pq -o r -- python $R/bench.py --steps 60 --warmup 8 --streams 1 --throughput-only > $R/gpurun_out/pq.log 2>&1
cd $R; python scratch/prof_summary.py gpurun_out/pq/r_results.db | head -14 | cut -c1-150
