set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q > $OUT/pytest_gpu_r02c_new.log 2>&1
echo "pytest(new) rc=$?"; tail -15 $OUT/pytest_gpu_r02c_new.log
export TMPDIR=/tmp; cd /tmp
for G in 1 2; do
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_g$G -- python3 $R/bench.py --groups $G --steps 200 --warmup 20 --min-timed-steps 200 --repeats 1 --no-extras --no-cpu-baseline --no-parity > $OUT/trace_g$G.log 2>&1
echo "trace G=$G rc=$?"
done
cd $R
for S in 0 1; do for G in 1 2; do
  QS_SPLIT=$S timeout -k 10 120 python bench.py --groups $G --no-extras --no-cpu-baseline --no-parity > $OUT/sweepS${S}_g${G}.json 2> $OUT/sweepS${S}_g${G}.err
  echo "QS_SPLIT=$S G=$G rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/sweepS${S}_g${G}.json')); print('%.3f G/s  period %.2f us  frac %.3f  timeline %.2f us' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac'], d['roofline']['gpu_timeline_us_per_step']))" 2>&1 | tail -1)"
done; done
