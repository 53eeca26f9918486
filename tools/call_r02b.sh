set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q -x > $OUT/pytest_gpu_r02b_new.log 2>&1
echo "pytest(new) rc=$?"; tail -15 $OUT/pytest_gpu_r02b_new.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_gpu_r02b_old.log 2>&1
echo "pytest(old) rc=$?"; tail -5 $OUT/pytest_gpu_r02b_old.log
for G in 1 2 3 4 8; do for TH in 1 0; do
  [ $G = 1 ] && [ $TH = 0 ] && continue
  timeout -k 10 120 python bench.py --groups $G --group-threads $TH --no-extras --no-cpu-baseline --no-parity > $OUT/sweep_g${G}_t${TH}.json 2> $OUT/sweep_g${G}_t${TH}.err
  echo "G=$G TH=$TH rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/sweep_g${G}_t${TH}.json')); print('%.3f G/s  period %.2f us  frac %.3f  timeline %.2f us' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac'], d['roofline']['gpu_timeline_us_per_step']))" 2>&1 | tail -1)"
done; done
