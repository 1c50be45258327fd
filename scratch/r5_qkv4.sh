#!/bin/bash
# round 5: p100 QKV projections - tile shapes of the LDS-DMA GEMM vs the row-block form, and what the phases cost
export VOG_PERF_EXPERIMENTS=1
echo -n "rowblock narrow: "; WL=cfg4 python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
echo -n "rowblock wide:   "; WL=cfg4 VOG_QKV_NARROW=0 python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
for t in 0 1 2 3 4 5 6 7; do
echo -n "tiled cfg $t: "; WL=cfg4 OPTS="qkv_lean=0" VOG_GEMM_TILE=$t python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
done
for d in 1 2 4 6; do
echo -n "tiled 128x64x2 debug $d: "; WL=cfg4 OPTS="qkv_lean=0" VOG_GEMM_TILE=4 VOG_GEMM_DEBUG=$d python scratch/mb_tail.py obj_qkv mul_pv 2>/dev/null | tail -1
done
