set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/runner_trace -- python3 $R/tools/runner_trace.py > $OUT/runner_trace.log 2>&1
echo "trace rc=$?"
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_r02a.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_r02a.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_r02a_20.json 2> $OUT/bench_r02a_20.err
echo "bench rc=$?"; head -c 600 $OUT/bench_r02a_20.json
