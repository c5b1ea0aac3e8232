#!/bin/bash
# probe builds of libmugiq_hip.so: one translation unit recompiled with -D<macro>=<n>, the others taken from the product build
# usage: tools/probes/build_exp.sh fused_mfma MUGIQ_MT_EXPERIMENT 1 2 3   -> tools/probes/build/libmugiq_hip_<macro>_<n>.so
set -e
R=$(cd $(dirname $0)/../.. && pwd)
TU=$1; MACRO=$2; shift 2
make -C $R/mugiq_amd/csrc -j4 > /dev/null
mkdir -p $R/tools/probes/build
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/mugiq_amd/csrc -Wall -Wno-unused-function -ffp-contract=fast -D$MACRO=$n \
    -c $R/mugiq_amd/csrc/$TU.hip -o $R/tools/probes/build/$TU.$MACRO.$n.o 2> /dev/null
  OBJS=$(ls $R/mugiq_amd/csrc/build/*.o | grep -v "/$TU.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/build/libmugiq_hip_${MACRO}_$n.so $OBJS $R/tools/probes/build/$TU.$MACRO.$n.o -ldl
  echo built $R/tools/probes/build/libmugiq_hip_${MACRO}_$n.so
done
