#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{ QUADSIM_HIP_LIB=/root/repo/quadsim_amd/csrc/libqs_exp_eidle.so timeout -k 10 200 python tools/runner_phases.py 65536 64 f32 2>&1 | grep -v amdgpu.ids | head -10; } > gpurun_out/runner_phases_exp.txt 2>&1
cat gpurun_out/runner_phases_exp.txt
