#!/bin/bash
# Diagnostic twin with in-kernel phase stamps (see QS_STAMP in quadsim_hip.hip); use with QUADSIM_HIP_LIB=...
set -e
cd "$(dirname "$0")/.."
hipcc -std=c++20 -O3 -DQS_STAMP -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -fPIC -shared -Wno-unused-result \
    quadsim_amd/csrc/quadsim_hip.hip -lhsa-runtime64 -o quadsim_amd/csrc/libquadsim_hip_stamp.so
echo built quadsim_amd/csrc/libquadsim_hip_stamp.so
