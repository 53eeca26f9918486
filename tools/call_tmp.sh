set -o pipefail
R=$(pwd); OUT=$R/gpurun_out
export TMPDIR=/tmp; cd /tmp
rm -rf $OUT/runner_trace4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/runner_trace4 -- python3 $R/tools/runner_trace.py > $OUT/runner_trace4.log 2>&1
echo "runner trace rc=$?"
cd $R
head -4 $OUT/runner_trace4/*/*kernel_stats.csv | cut -c1-150
