#!/bin/bash
# build_variant.sh NAME "-DFLAG=.. ..." : libvog_hip variant under scratch/tmp/NAME/ (only the TUs that see the flags are rebuilt)
set -e
N=$1; F=$2; C=vognet-pytorch_amd/csrc; O=scratch/tmp/$N; mkdir -p $O
python $C/build.py >/dev/null
TUS=${3:-"lstm pair visenc"}
for t in $TUS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $F -c $C/$t.hip -o $O/$t.o & done; wait
OBJS=""; for s in forward gemm attention elementwise lstm txtail visenc pair loss assemble backward; do if [ -f $O/$s.o ]; then OBJS="$OBJS $O/$s.o"; else OBJS="$OBJS $C/$s.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libvog_hip.so $OBJS
ls -la $O/libvog_hip.so
