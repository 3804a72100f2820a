#!/bin/bash
# instrumented builds of libkf_hip.so (same ABI) for the tape-path experiments: gpurun_out-free, built in-tree
set -e
cd "$(dirname "$0")/../../pymc_statespace_amd/csrc"
OBJS=$(ls *.o | grep -v kf_forward_b4.o)
for V in ${VARIANTS:-NO_GSTORE NO_PTAPE NO_ATAPE}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast $(echo $V | sed 's/+/ -DB4_/g; s/^/-DB4_/') -c kf_forward_b4.hip -o /tmp/b4_$V.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkf_hip_$V.so $OBJS /tmp/b4_$V.o
done
ls -la ../libkf_hip_*.so
