set -o pipefail
R=$(pwd); OUT=$R/gpurun_out
echo "== tests, default"; timeout -k 10 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
echo "== private-queue tests with QS_CHAIN_OVERLAP=1"; QS_CHAIN_OVERLAP=1 timeout -k 10 300 python -m pytest tests/test_gpu_groups_and_rollout.py -m gpu -q -x -k "private" 2>&1 | tail -4
run() { timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --no-parity "$@" > $OUT/tmp.json 2> $OUT/tmp.err; python -c "import json; d=json.load(open('$OUT/tmp.json')); print('%.3f G/s  period %.2f us  frac %.3f  [%s x%s]' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac'], d['config']['queue_mode'], d['config']['private_queues']))" 2>&1 | tail -1; }
for Q in 1 2; do
echo "barrier-bit chain x$Q: $(run --queue-mode private --queues $Q)"
echo "overlapped chain  x$Q: $(QS_CHAIN_OVERLAP=1 run --queue-mode private --queues $Q)"
done
echo "overlapped x1 pool16: $(QS_CHAIN_OVERLAP=1 run --queue-mode private --queues 1 --action-pool 16)"
echo "overlapped x1 131072: $(QS_CHAIN_OVERLAP=1 run --queue-mode private --queues 1 --envs-per-gpu 131072)"
