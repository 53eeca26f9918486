#!/bin/bash
# builds the two artefacts of the HSA-queue experiment (tools/hsa_chain_exp.py): the device code object and the dispatcher
set -e
cd "$(dirname "$0")/.."
hipcc -std=c++20 -O3 -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 --cuda-device-only quadsim_amd/csrc/quadsim_hip.hip -o tools/quadsim_dev.bundle
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=tools/quadsim_dev.bundle --output=tools/quadsim_dev.hsaco
rm -f tools/quadsim_dev.bundle
g++ -O2 -fPIC -shared tools/hsa_chain_exp.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o tools/libqs_hsa_exp.so
echo built tools/quadsim_dev.hsaco tools/libqs_hsa_exp.so
