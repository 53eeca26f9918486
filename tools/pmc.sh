#!/bin/bash
# HBM-side traffic of the step kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), at the bench size and at a size far beyond L2 + Infinity Cache for calibration.
TAG=${1:-r01}
R=$(pwd); OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for N in 65536 2097152; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${C}_$N -- python3 $R/bench.py --envs-per-gpu $N --steps 200 --warmup 20 --action-pool 8 --no-cpu-baseline --no-extras --no-parity > $OUT/${C}_$N.log 2>&1 || echo "pass $C $N failed"
  done
done
cd $R
python tools/pmc_summary.py $OUT/FETCH_SIZE_65536 $OUT/WRITE_SIZE_65536 $OUT/FETCH_SIZE_2097152 $OUT/WRITE_SIZE_2097152 | grep -E "k_env|fill_actions" | tee $OUT/summary.txt
