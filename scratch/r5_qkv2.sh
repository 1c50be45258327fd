#!/bin/bash
# round 5: what the cfg-2 QKV GEMMs spend their time on (VOG_GEMM_DEBUG: 1 no DMA, 2 no MFMA, 4 no epilogue; wrong results)
export VOG_PERF_EXPERIMENTS=1
for d in 0 1 2 4 6 7; do
echo -n "debug $d: "; OPTS="pair_launches=0" VOG_GEMM_DEBUG=$d python scratch/mb_tail.py obj_qkv mul_pv mul_pl 2>/dev/null | tail -1
done
for t in 5 7 2 4 1; do
echo -n "tile $t: "; OPTS="pair_launches=0" VOG_GEMM_TILE=$t python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
done
