#!/bin/bash
# Debug twin of the library: -DQS_DEBUG turns every QS_ASSERT into an in-kernel assert (bound checks of the global
# indices of the env / roll-out kernels).  Use with QUADSIM_HIP_LIB=quadsim_amd/csrc/libquadsim_hip_dbg.so.
set -e
cd "$(dirname "$0")/.."
hipcc -std=c++20 -O3 -DQS_DEBUG -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -fPIC -shared -Wno-unused-result \
    quadsim_amd/csrc/quadsim_hip.hip -lhsa-runtime64 -o quadsim_amd/csrc/libquadsim_hip_dbg.so
echo built quadsim_amd/csrc/libquadsim_hip_dbg.so
