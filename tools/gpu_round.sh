#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprofv3 kernel stats.  Run via gpurun from the repo root.
# Usage: bash tools/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
REPO=$(pwd)
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -3 $OUT/pytest_gpu_$TAG.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke_$TAG.log
timeout -k 10 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$?"; cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
cd $REPO
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r cat | head -12
