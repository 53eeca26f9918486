#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{ timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "fast or bf16 or runner or policy or Runner" 2>&1 | tail -3
for i in 1 2; do for L in libqs_prev.so libquadsim_hip.so; do
  QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/$L timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1
  QUADSIM_RUNNER_SERIAL=1 QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/$L timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1
done; done; } > gpurun_out/ab_weave.txt 2>&1
cat gpurun_out/ab_weave.txt
