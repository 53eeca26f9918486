#!/bin/bash
# HBM-side traffic of the step kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), at the bench size in both queue modes and at a size far beyond L2 + Infinity Cache for calibration.
TAG=${1:-r01}
R=$(pwd); OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
X="--steps 200 --warmup 20 --min-timed-steps 200 --repeats 1 --action-pool 8 --no-cpu-baseline --no-extras --no-parity"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${C}_65536 -- python3 $R/bench.py --envs-per-gpu 65536 --queue-mode hip $X > $OUT/${C}_65536.log 2>&1 || echo "pass $C hip failed"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${C}_65536_private -- python3 $R/bench.py --envs-per-gpu 65536 --queue-mode private --queues 1 $X > $OUT/${C}_65536_private.log 2>&1 || echo "pass $C private failed"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${C}_2097152 -- python3 $R/bench.py --envs-per-gpu 2097152 --queue-mode hip $X > $OUT/${C}_2097152.log 2>&1 || echo "pass $C 2M failed"
done
cd $R
python tools/pmc_summary.py $OUT/FETCH_SIZE_65536 $OUT/WRITE_SIZE_65536 $OUT/FETCH_SIZE_65536_private $OUT/WRITE_SIZE_65536_private $OUT/FETCH_SIZE_2097152 $OUT/WRITE_SIZE_2097152 | grep -E "k_env|fill_actions" | tee $OUT/summary.txt
