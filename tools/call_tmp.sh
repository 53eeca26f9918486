R=$(pwd); OUT=$R/gpurun_out
run() { QUADSIM_HIP_LIB=$R/$1 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --no-parity ${@:2} > $OUT/tmp.json 2> $OUT/tmp.err; python -c "import json; d=json.load(open('$OUT/tmp.json')); print('%.3f G/s  period %.2f us  frac %.3f' % (d['value']/1e9, d['roofline']['step_period_us'], d['roofline']['frac']))" 2>&1 | tail -1; }
NAMES=("plain" "nt" "sc0" "sc1" "sc0 sc1" "sc0 nt" "sc1 nt" "sc0 sc1 nt")
echo "builtin-nt (shipping)      : $(run quadsim_amd/csrc/libquadsim_hip.so)"
for i in 0 1 2 3 4 5 6 7; do echo "asm [${NAMES[$i]}] 65536: $(run quadsim_amd/csrc/libqs_ab_$i.so)   | 1M: $(run quadsim_amd/csrc/libqs_ab_$i.so --envs-per-gpu 1048576 --steps 300 --warmup 30 --min-timed-steps 300)"; done
echo "builtin-nt (shipping)      : $(run quadsim_amd/csrc/libquadsim_hip.so)"
