#!/bin/bash
# round 5: the BiLSTM input projections of cfg 3 / cfg 5 (Bn*T > 64: separate GEMM launches): tile shapes
export VOG_PERF_EXPERIMENTS=1
for wl in cfg3 cfg5; do
echo -n "$wl default: "; WL=$wl python scratch/mb_tail.py lstm_ih0 lstm_ih1 2>/dev/null | tail -1
for t in 0 1 2 3 4 5 6 7; do
echo -n "$wl tile $t: "; WL=$wl VOG_GEMM_TILE=$t python scratch/mb_tail.py lstm_ih0 lstm_ih1 2>/dev/null | tail -1
done; done
