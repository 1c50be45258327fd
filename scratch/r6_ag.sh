#!/bin/bash
# UNSAFE experiment: more than 4 forwards in flight (no guard on the number of co-resident persistent BiLSTM kernels) - what would it buy?
B="timeout 120 python bench.py --steps 400 --warmup 40 --throughput-only --no-cpu-baseline"
echo "4 streams, 4 queues: $($B 2>&1 | tail -1)"
export VOG_PERF_EXPERIMENTS=1 VOG_LSTM_PERSISTENT=1
for q in 8; do for n in 4 5 6 8; do
  echo "queues $q inflight $n: $(GPU_MAX_HW_QUEUES=$q VOG_MAX_INFLIGHT=$n $B --streams $n 2>&1 | tail -1 | cut -c1-200)"
done; done
echo "queues 4 inflight 6: $(VOG_MAX_INFLIGHT=6 $B --streams 6 2>&1 | tail -1 | cut -c1-200)"
echo "queues 4 inflight 8: $(VOG_MAX_INFLIGHT=8 $B --streams 8 2>&1 | tail -1 | cut -c1-200)"
