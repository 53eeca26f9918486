set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/call_profiles_r02.sh 2>&1 | tail -6
bash tools/call_bench_r02.sh 2>&1 | tail -14
