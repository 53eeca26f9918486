set -o pipefail
R=$(pwd); OUT=$R/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_t.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_t.log
run() { QUADSIM_HIP_LIB=$R/$1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --no-parity ${@:2} > $OUT/tmp.json 2> $OUT/tmp.err; python -c "import json; d=json.load(open('$OUT/tmp.json')); print('%.3f G/s  period %.2f us  frac %.3f  [%s x%s]' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac'], d['config']['queue_mode'], d['config']['private_queues']))" 2>&1 | tail -1; }
A=quadsim_amd/csrc/libquadsim_hip_prev.so; B=quadsim_amd/csrc/libquadsim_hip.so
for i in 1 2; do
echo "prev private x2: $(run $A --queue-mode private --queues 2)"; echo "new  private x2: $(run $B --queue-mode private --queues 2)"
echo "prev private x1: $(run $A --queue-mode private --queues 1)"; echo "new  private x1: $(run $B --queue-mode private --queues 1)"
echo "prev hip       : $(run $A --queue-mode hip)"; echo "new  hip       : $(run $B --queue-mode hip)"
done
echo "new private x2 pool 16: $(run $B --queue-mode private --queues 2 --action-pool 16)"
