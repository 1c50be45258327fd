#!/bin/bash
for i in 1 2; do ./scratch/ts_attn6; done; echo direct-store; for i in 1 2; do ./scratch/ts_attn6_direct; done
bash scratch/r6_n.sh
