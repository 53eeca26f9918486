#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_tests.txt; cat gpurun_out/gpu_tests.txt
