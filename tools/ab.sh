#!/bin/bash
# quick A/B of library builds: bash tools/ab.sh libA.so libB.so ...  (bench step-API only, interleaved rounds)
for round in 1 2 3; do
for lib in "$@"; do
  QUADSIM_HIP_LIB=$(pwd)/$lib python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'G env-steps/s %.3f'%(d['value']/1e9), 'launch_us %.2f'%d['roofline']['launch_period_us'])"
done; done
