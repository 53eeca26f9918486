#!/bin/bash
# interleaved A/B of library builds on the step API: bash tools/ab_libs.sh libA.so libB.so [rounds] [extra bench args]
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do for L in $A $B; do
  QUADSIM_HIP_LIB=$R/$L timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-parity "$@" > $OUT/ab_tmp.json 2> $OUT/ab_tmp.err
  echo "$L $(python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('%.3f G/s  period %.2f us  frac %.3f' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac']))" 2>&1 | tail -1)"
done; done
