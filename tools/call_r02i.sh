set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_dbg.so timeout -k 10 600 python tools/soak_shape_debug.py > $OUT/soak_shape_debug.log 2>&1
echo "debug soak rc=$?"; tail -12 $OUT/soak_shape_debug.log
QUADSIM_HIP_LIB=$R/quadsim_amd/csrc/libquadsim_hip_dbg.so timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_dbg.log 2>&1
echo "pytest (debug-assert library) rc=$?"; tail -4 $OUT/pytest_gpu_dbg.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
echo "self-launched 2-rank gloo rehearsal rc=$?"; head -c 700 $OUT/bench_gloo2.json; echo; tail -3 $OUT/bench_gloo2.err
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/runner_trace2 -- python3 $R/tools/runner_trace.py > $OUT/runner_trace2.log 2>&1
echo "runner trace rc=$?"
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_r02i_20.json 2> $OUT/bench_r02i_20.err
echo "bench 20 rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench_r02i_20.json'))
print({k: d[k] for k in ('value','ms_per_step','steps')}); print(d['roofline']); print(d['parity']); print(d.get('config1')); print(d.get('cpu_baseline'))"
