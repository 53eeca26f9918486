#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{ timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 &&
echo "--- QS_DEBUG library (device asserts), Runner / roll-out tests" &&
QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/libquadsim_hip_dbg.so timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "runner or Runner or rollout or policy" 2>&1 | tail -3 &&
echo "--- soak: Runner.run() x 60 at 65 536 x 600" &&
timeout -k 10 400 python tools/soak_runner.py 2>&1 | grep -v "issued" | tail -5 &&
for i in 1 2; do timeout -k 10 200 python tools/ab_policy.py 2>&1 | tail -1; done; } > gpurun_out/final_check.txt 2>&1
cat gpurun_out/final_check.txt
