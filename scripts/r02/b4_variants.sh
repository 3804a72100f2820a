#!/bin/bash
# instrumented builds of libkf_hip.so (same ABI) for the tape-path experiments: gpurun_out-free, built in-tree
set -e
cd "$(dirname "$0")/../../pymc_statespace_amd/csrc"
SRC=${SRC:-kf_forward_b4}
PFX=${PFX:-B4_}
OBJS=$(ls *.o | grep -v "^$SRC.o")
for V in ${VARIANTS:-NO_GSTORE NO_PTAPE NO_ATAPE}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast $(echo $V | sed "s/+/ -D$PFX/g; s/^/-D$PFX/") -c $SRC.hip -o /tmp/b4_$V.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkf_hip_$V.so $OBJS /tmp/b4_$V.o
done
ls -la ../libkf_hip_*.so
