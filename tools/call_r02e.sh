set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
for RND in 1 0 1 0; do
  timeout -k 10 120 python bench.py --randomise $RND --no-extras --no-cpu-baseline --no-parity > $OUT/rnd_$RND.json 2> $OUT/rnd_$RND.err
  echo "randomise=$RND rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/rnd_$RND.json')); print('%.3f G/s  period %.2f us  frac %.3f' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac']))" 2>&1 | tail -1)"
done
export QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_stamp.so
