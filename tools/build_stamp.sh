#!/bin/bash
# Diagnostic twins with in-kernel stamps (see QS_STAMP in quadsim_hip.hip); use with QUADSIM_HIP_LIB=...
#   libquadsim_hip_stamp.so   -DQS_STAMP      every phase boundary of the step kernel (costs ~0.7 us per step)
#   libquadsim_hip_stamp2.so  -DQS_STAMP=2    first / last stamp of each wave only: period, span and gap of the unperturbed chain
set -e
cd "$(dirname "$0")/.."
F="-std=c++20 -O3 -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -fPIC -shared -Wno-unused-result"
hipcc $F -DQS_STAMP quadsim_amd/csrc/quadsim_hip.hip -lhsa-runtime64 -o quadsim_amd/csrc/libquadsim_hip_stamp.so &
hipcc $F -DQS_STAMP=2 quadsim_amd/csrc/quadsim_hip.hip -lhsa-runtime64 -o quadsim_amd/csrc/libquadsim_hip_stamp2.so &
wait
echo built quadsim_amd/csrc/libquadsim_hip_stamp.so quadsim_amd/csrc/libquadsim_hip_stamp2.so
