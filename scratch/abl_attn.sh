#!/bin/bash
# ablation of attn_tile2 (wrong results): v1 = no softmax work in the loop, v2 = no MFMAs in the loop
cp vognet-pytorch_amd/csrc/libvog_hip.so /tmp/good.so
echo "full: $(python scratch/mb_attn_long.py 20 1)"
for v in v1 v2; do cp scratch/libvog_$v.so vognet-pytorch_amd/csrc/libvog_hip.so; echo "$v: $(python scratch/mb_attn_long.py 20 1)"; done
cp /tmp/good.so vognet-pytorch_amd/csrc/libvog_hip.so
