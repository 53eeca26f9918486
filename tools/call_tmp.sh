#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{ for L in libqs_pf1.so libqs_pf2.so libqs_pf3.so; do echo "=== $L"; QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/$L timeout -k 10 200 python tools/runner_phases.py 65536 64 bf16x3 2>&1 | grep -v amdgpu.ids || exit 1;
  QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/$L timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1; QUADSIM_RUNNER_SERIAL=1 QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/$L timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1; done; } > gpurun_out/runner_phases_exp.txt 2>&1
cat gpurun_out/runner_phases_exp.txt
