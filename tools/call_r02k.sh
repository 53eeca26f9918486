R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-parity "$@" > $OUT/tmp.json 2> $OUT/tmp.err; python -c "import json; d=json.load(open('$OUT/tmp.json')); print('%.3f G/s  period %.2f us  frac %.3f' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac']))" 2>&1 | tail -1; }
echo "baseline $(run)"
for BIT in 0 3 5 8; do for ST in 2 4 6 8 12; do
  echo "bit=$BIT stagger=$ST $(QS_STAGGER=$ST QS_STAGGER_BIT=$BIT run)"
done; done
echo "baseline $(run)"
