#!/bin/bash
# round 4, experiment 1: does the LDS footprint of the persistent BiLSTM layer launch decide what the other streams'
# kernels can do on its 64 CUs? unpaired launches (every body has its own LDS size), LSTM LDS = 78.5 KB (new: gates
# sized by the real column count) / +10 KB (= round 3's 88.5 KB) / padded to 156 KB (exclusive CU)
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4lds; mkdir -p $O
run() { python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
echo "paired   lstm_lds=default        -> $(run)"
echo "unpaired lstm_lds=78.5K          -> $(run --set pair_launches=0)"
echo "unpaired lstm_lds=88.5K          -> $(VOG_LSTM_LDS_EXTRA=10240 run --set pair_launches=0)"
echo "unpaired lstm_lds=156K           -> $(VOG_LSTM_LDS_EXTRA=79000 run --set pair_launches=0)"
echo "unpaired lean lstm_lds=78.5K   -> $(run --set pair_launches=0 --set enc_lean=1)"
echo "unpaired lean lstm_lds=156K    -> $(VOG_LSTM_LDS_EXTRA=79000 run --set pair_launches=0 --set enc_lean=1)"
done
} > $O/lds.log 2>&1
cat $O/lds.log
