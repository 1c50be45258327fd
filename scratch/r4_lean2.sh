#!/bin/bash
# round 4: lean (<= 128 VGPR) attention kernels of both transformers, pair_mask, qkv_lean: whole chip and half of it
export VOG_PERF_EXPERIMENTS=1
O=gpurun_out/r4lean2; mkdir -p $O
run() { timeout 200 python bench.py --steps 800 --warmup 80 --throughput-only "$@" 2>/dev/null | tail -1; }
{
for rep in 1 2; do
for cu in 255 127; do
M="HSA_CU_MASK=0:0-$cu"
echo "CUs 0-$cu: both lean (default)          -> $(env $M bash -c "$(declare -f run); run")"
echo "CUs 0-$cu: old frag, lean struct1       -> $(env $M VOG_ATTN_FRAG_LEAN=0 bash -c "$(declare -f run); run")"
echo "CUs 0-$cu: both old                     -> $(env $M VOG_ATTN_FRAG_LEAN=0 VOG_ATTN_STRUCT1_LEAN=0 bash -c "$(declare -f run); run")"
echo "CUs 0-$cu: both lean + pair_mask=3      -> $(env $M bash -c "$(declare -f run); run --set pair_mask=3")"
echo "CUs 0-$cu: both lean + pair_mask=3 + qkv_lean -> $(env $M bash -c "$(declare -f run); run --set pair_mask=3 --set qkv_lean=1")"
echo "CUs 0-$cu: both lean + pair_mask=1      -> $(env $M bash -c "$(declare -f run); run --set pair_mask=1")"
done
done
echo "streams=1: lean $(run --streams 1) old $(VOG_ATTN_STRUCT1_LEAN=0 VOG_ATTN_FRAG_LEAN=0 run --streams 1)"
} > $O/lean2.log 2>&1
cat $O/lean2.log
